// Light-weight stand-ins for the third-party types that appear in the reference's feature/ and
// matching/ interfaces (cv::Mat, cv::Point2f, Eigen::Vector2f, GeometricCamera), so that the
// host-side mirror of PPGExtractor / Matcher compiles in a tree without OpenCV and Eigen.
// A host project that has those libraries converts at this boundary; field
// names and meanings follow the reference:
//   KeyPointEx  sensors/include/GeometricCamera.h:22-37
//   KeyEdge     feature/include/PPGGraph.h:32-55, feature/src/PPGGraph.cpp:16-32
#pragma once
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace ppg_host {

struct Vec2f {  // Eigen::Vector2f
    float v[2] = {0.f, 0.f};
    Vec2f() {}
    Vec2f(float x, float y) { v[0] = x; v[1] = y; }
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
    float x() const { return v[0]; }
    float y() const { return v[1]; }
};

struct Point2f {  // cv::Point2f
    float x = 0.f, y = 0.f;
    Point2f() {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

class KeyPointEx {
public:
    KeyPointEx() {}
    KeyPointEx(float x_, float y_, float sc) : mPos(x_, y_), mfScore(sc), mbOut(true) {}
    Vec2f mPos, mPosUn;
    float mfScore = 0.f;
    std::vector<unsigned int> mvConnected;
    std::vector<std::pair<unsigned int, unsigned int>> mvColine;
    bool mbOut = true;
};

class KeyEdge {
public:
    KeyEdge() : startIdx(0), endIdx(0), isBad(true), lscore(0.f), length(0.f) {}
    KeyEdge(const unsigned int& i0, const unsigned int& i1) : startIdx(i0), endIdx(i1), isBad(false), lscore(0.f), length(0.f) {}
    unsigned int theOtherPid(const unsigned int pid) const { return pid == startIdx ? endIdx : startIdx; }
    unsigned int startIdx, endIdx;
    bool isBad;
    float lscore;
    float length;  // never written on this path (as in the reference)
};

// cv::Mat of type CV_8UC1, borrowed for the duration of a call
struct GrayImage {
    const uint8_t* data = nullptr;
    int rows = 0, cols = 0, step = 0;  // step = bytes per row
    int channels() const { return 1; }
};

// cv::Mat of type CV_32FC1, owning
struct DescMat {
    int rows = 0, cols = 0;
    std::vector<float> data;
    DescMat() {}
    DescMat(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0.f) {}
    float* ptr(int r) { return data.data() + (size_t)r * cols; }
    const float* ptr(int r) const { return data.data() + (size_t)r * cols; }
    DescMat row(int r) const {
        DescMat m(1, cols);
        std::memcpy(m.data.data(), ptr(r), sizeof(float) * cols);
        return m;
    }
};

// What PPGExtractor and Matcher read from GeometricCamera (feature/src/PPGExtractor.cpp:58-62,
// sensors/src/GeometricCamera.cpp:26-61): toK(), toD(), imWidth(), imHeight(), mnType.
struct CameraModel {
    static const unsigned int CAM_PINHOLE = 0, CAM_FISHEYE = 1;
    float K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float D[4] = {0, 0, 0, 0};
    int width = 0, height = 0;
    unsigned int mnType = CAM_PINHOLE;
    int imWidth() const { return width; }
    int imHeight() const { return height; }
};

}  // namespace ppg_host
