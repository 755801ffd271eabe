// Host-side mirror of the reference's PPGExtractor (feature/include/PPGExtractor.h:34-148) over the
// C ABI of libppgfront.  Same method names, argument meaning and copy semantics; the GPU work happens
// behind ppg_extract*.  Header-only; link with -lppgfront.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ppgfront.h"
#include "ppg_types.h"

namespace ppg_host {

class PPGExtractor {
public:
    // feature/include/PPGExtractor.h:42 — dataPath is the directory holding ppg_weights.bin (the fp32
    // parameters of Backbone/PointHeatmap/EdgeHeatmap/Descriptor.pt, PPGExtractor.cpp:77-80).
    PPGExtractor(const CameraModel* pCam, std::string dataPath, int device_id = 0) : mCam(*pCam), mDataPath(std::move(dataPath)), mDevice(device_id) {
        mbFisheye = (pCam->mnType == CameraModel::CAM_FISHEYE);
        rebuild();
    }
    ~PPGExtractor() { ppg_destroy(mCtx); }
    PPGExtractor(const PPGExtractor&) = delete;
    PPGExtractor& operator=(const PPGExtractor&) = delete;

    // PPGExtractor::run (feature/src/PPGExtractor.cpp:118-147)
    void run(GrayImage srcMat, std::vector<KeyPointEx>& _keypoints, std::vector<KeyPointEx>& _keypoints_un,
             std::vector<KeyEdge>& _keyedges, DescMat& _descriptors) {
        if (srcMat.channels() != 1 || srcMat.rows != mCam.height || srcMat.cols != mCam.width)
            throw std::invalid_argument("PPGExtractor::run: image must be single-channel and match the camera size");
        apply_tunables();
        ppg_frame_out o;
        check(ppg_extract(mCtx, srcMat.data, srcMat.step, &o));
        absorb(o);
        _keypoints = mvKeyPoints;
        _keyedges = mvKeyEdges;
        const unsigned int pointSize = (unsigned int)mvKeyPoints.size();
        _descriptors = DescMat((int)pointSize, DESC_DIM_SIZE);
        if (pointSize) std::memcpy(_descriptors.data.data(), normDesc.data.data(), sizeof(float) * pointSize * DESC_DIM_SIZE);
        if (!mbFisheye)
            for (unsigned int jid = 0; jid < pointSize; jid++) _keypoints[jid].mPos = _keypoints[jid].mPosUn;
        _keypoints_un = _keypoints;
    }

    // The four public pipeline stages (feature/include/PPGExtractor.h:56-65)
    void inference(GrayImage undistortedMat) {
        apply_tunables();
        std::vector<uint8_t> tight((size_t)mCam.width * mCam.height);
        for (int y = 0; y < mCam.height; y++)
            std::memcpy(tight.data() + (size_t)y * mCam.width, undistortedMat.data + (size_t)y * undistortedMat.step, mCam.width);
        check(ppg_inference(mCtx, tight.data(), 1, 0));
    }
    void detectKeyPoint() { check(ppg_detect_keypoints(mCtx, 1)); }
    void detectLines() { check(ppg_detect_lines(mCtx, 1)); }
    void genPointDescriptor() {
        check(ppg_gen_descriptors(mCtx, 1));
        ppg_frame_out o;
        check(ppg_collect(mCtx, 1, &o));
        absorb(o);
    }

    const std::vector<KeyPointEx>& getKeyPoints() const { return mvKeyPoints; }
    const std::vector<KeyEdge>& getKeyEdges() const { return mvKeyEdges; }
    ppg_ctx* context() { return mCtx; }

public:
    std::vector<KeyPointEx> mvKeyPoints;
    std::vector<KeyEdge> mvKeyEdges;
    DescMat normDesc;

    // feature/include/PPGExtractor.h:105-114, defaults feature/src/PPGExtractor.cpp:44-53
    inline static int DESC_DIM_SIZE = 256;
    inline static float JUNCTION_THRESH = 1.0f / 128.0f;
    inline static int JUNCTION_NMS_RADIUS = 4;
    inline static unsigned int JUNCTION_MAX_NUM = 500;
    inline static float LINE_VALID_THRESH = 1.0e-2f;
    inline static float LINE_VALID_RATIO = 0.3f;
    inline static float LINE_DISTTHRESH = 2.0f;
    inline static int HEATMAP_REFINE_SZ = 16;
    inline static float LINE_HEATMAP_THRESH = 0.2f;
    inline static float LINE_INLIER_RATE = 0.8f;

private:
    void check(int rc) {
        if (rc != PPG_OK) throw std::runtime_error(std::string("libppgfront: ") + ppg_last_error(mCtx));
    }
    void fill_cfg(ppg_config& c) const {
        ppg_default_config(&c, mCam.width, mCam.height);
        std::memcpy(c.K, mCam.K, sizeof(c.K));
        std::memcpy(c.D, mCam.D, sizeof(c.D));
        c.fisheye = mbFisheye ? 1 : 0;
        c.device_id = mDevice;
        c.desc_dim = DESC_DIM_SIZE; c.junction_thresh = JUNCTION_THRESH; c.junction_nms_radius = JUNCTION_NMS_RADIUS;
        c.junction_max_num = JUNCTION_MAX_NUM; c.line_valid_thresh = LINE_VALID_THRESH; c.line_valid_ratio = LINE_VALID_RATIO;
        c.line_dist_thresh = LINE_DISTTHRESH; c.heatmap_refine_sz = HEATMAP_REFINE_SZ;
        c.line_heatmap_thresh = LINE_HEATMAP_THRESH; c.line_inlier_rate = LINE_INLIER_RATE;
    }
    bool tunables_changed() const {
        ppg_config c;
        fill_cfg(c);
        return c.desc_dim != mCfg.desc_dim || c.junction_thresh != mCfg.junction_thresh ||
               c.junction_nms_radius != mCfg.junction_nms_radius || c.junction_max_num != mCfg.junction_max_num ||
               c.line_valid_thresh != mCfg.line_valid_thresh || c.line_valid_ratio != mCfg.line_valid_ratio ||
               c.line_dist_thresh != mCfg.line_dist_thresh || c.heatmap_refine_sz != mCfg.heatmap_refine_sz ||
               c.line_heatmap_thresh != mCfg.line_heatmap_thresh || c.line_inlier_rate != mCfg.line_inlier_rate;
    }
    // The reference's tunables are mutable statics read on every frame (feature/include/PPGExtractor.h:105-114).  The
    // thresholds are handed to the context as they are (ppg_set_tunables: no buffer, weight or warm-up is touched); only
    // the three that size buffers and kernels (DESC_DIM_SIZE, JUNCTION_NMS_RADIUS, JUNCTION_MAX_NUM) re-create it.
    void apply_tunables() {
        if (!tunables_changed()) return;
        ppg_config c = mCfg;
        fill_cfg(c);
        c.weights_path = mCfg.weights_path;
        c.max_batch = mCfg.max_batch;
        const int rc = ppg_set_tunables(mCtx, &c);
        if (rc == PPG_OK) { mCfg = c; return; }
        if (rc != PPG_ECAP) check(rc);
        rebuild();
    }
    void rebuild() {
        if (mCtx) ppg_destroy(mCtx);
        mWeights = mDataPath + "/ppg_weights.bin";
        fill_cfg(mCfg);
        mCfg.weights_path = mWeights.c_str();
        mCfg.max_batch = 1;
        mCtx = ppg_create(&mCfg);
        if (!mCtx) throw std::runtime_error(std::string("ppg_create: ") + ppg_last_error(nullptr));
    }
    void absorb(const ppg_frame_out& o) {
        mvKeyPoints.clear();
        mvKeyPoints.reserve(JUNCTION_MAX_NUM);
        for (int i = 0; i < o.n; i++) {
            KeyPointEx kp((float)o.kp_xy[2 * i], (float)o.kp_xy[2 * i + 1], o.kp_score[i]);
            kp.mPosUn = Vec2f(o.kp_un[2 * i], o.kp_un[2 * i + 1]);
            kp.mbOut = o.kp_out[i] != 0;
            for (int k = o.conn_off[i]; k < o.conn_off[i + 1]; k++) kp.mvConnected.push_back(o.conn_idx[k]);
            for (int k = o.coline_off[i]; k < o.coline_off[i + 1]; k++)
                kp.mvColine.emplace_back(o.coline_pairs[2 * k], o.coline_pairs[2 * k + 1]);
            mvKeyPoints.push_back(std::move(kp));
        }
        // With no key points the reference returns before touching mvKeyEdges (PPGExtractor.cpp:239-240):
        // the edges of the previous frame stay in the member.  Preserved here.
        if (o.n > 0) {
            mvKeyEdges.clear();
            for (int e = 0; e < o.n_edges; e++) {
                KeyEdge ke(o.edge_idx[2 * e], o.edge_idx[2 * e + 1]);
                ke.lscore = o.edge_score[e];
                mvKeyEdges.push_back(ke);
            }
        }
        normDesc = DescMat(o.n, DESC_DIM_SIZE);
        if (o.n) std::memcpy(normDesc.data.data(), o.desc, sizeof(float) * o.n * DESC_DIM_SIZE);
    }

    CameraModel mCam;
    std::string mDataPath, mWeights;
    int mDevice = 0;
    bool mbFisheye = false;
    ppg_config mCfg;
    ppg_ctx* mCtx = nullptr;
};

}  // namespace ppg_host
