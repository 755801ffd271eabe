// Worker threads for the host-side part of a batch.  Pure C++ (no HIP): included by ppg_front.cpp and by the
// stress test tests/host_pool_stress.cpp.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// Host worker threads of a context: run(n, fn) calls fn(0) .. fn(n-1), each index once, on the workers and the
// calling thread, indices handed out one at a time, and returns when all are done.  The threads live as long as the
// pool (no spawn per batch).
class HostPool {
public:
    explicit HostPool(int workers) {
        for (int t = 0; t < workers; t++) th_.emplace_back([this]() { loop(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void run(int n, const std::function<void(int)>& fn) {
        uint64_t gen;
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; n_ = n; left_ = n;
            gen = ++gen_;
            ticket_.store(gen << 32);  // (job generation, next index): a thread still holding an older job cannot claim
        }
        cv_.notify_all();
        work(&fn, n, gen);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return left_ == 0; });
        fn_ = nullptr;
    }

private:
    void work(const std::function<void(int)>* fn, int n, uint64_t gen) {
        int did = 0;
        uint64_t v = ticket_.load();
        for (;;) {
            if ((v >> 32) != gen || (int)(v & 0xffffffffu) >= n) break;
            if (!ticket_.compare_exchange_weak(v, v + 1)) continue;
            (*fn)((int)(v & 0xffffffffu));
            did++;
            v = ticket_.load();
        }
        if (did) {
            // an index is claimed only while its job is current, and run() returns only after every claimed index
            // has been counted here, so fn is alive whenever it is called
            std::lock_guard<std::mutex> lk(m_);
            left_ -= did;
            if (left_ == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* fn;
            int n;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return stop_ || (gen_ != seen && fn_); });
                if (stop_) return;
                seen = gen_; fn = fn_; n = n_;
            }
            work(fn, n, seen);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<uint64_t> ticket_{0};
    int n_ = 0, left_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

