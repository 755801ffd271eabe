/*
 * libppgfront — C ABI of the MI355X-native PPG-SLAM feature front-end.
 *
 * Drop-in boundary for the reference's feature/ (PPGExtractor) and matching/ (Matcher,
 * DescriptorDistance) interfaces.  Plain pointers and sizes only; no C++ or torch types.
 * Every call returns 0 on success or a negative PPG_E* code; ppg_last_error() gives the text.
 * Nothing throws across this boundary.  A context is bound to one GPU and is not thread-safe
 * (one context per stream / host thread), like the reference's single PPGExtractor instance
 * (system/src/Tracking.cpp:51).
 *
 * Citations are relative to the reference repository root.
 */
#ifndef PPGFRONT_H
#define PPGFRONT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPG_OK 0
#define PPG_EINVAL (-1)   /* bad argument / shape */
#define PPG_EHIP (-2)     /* HIP runtime error */
#define PPG_EIO (-3)      /* weight blob unreadable or malformed */
#define PPG_ENOMEM (-4)
#define PPG_ECAP (-5)     /* a capacity given at create time was exceeded */
#define PPG_ERANGE (-6)   /* precision 3: an activation left the range of the fp16 split (use precision 0 or 1) */
#define PPG_ECOMM (-7)    /* RCCL could not be loaded, or one of its calls failed (multi-GPU mode) */

typedef struct ppg_ctx ppg_ctx;

/* Replaces the PPGExtractor constructor arguments (feature/include/PPGExtractor.h:42: camera K, D,
 * size, fisheye flag read at feature/src/PPGExtractor.cpp:58-62; net directory :77-80) and the ten
 * public static tunables (feature/include/PPGExtractor.h:105-114, defaults PPGExtractor.cpp:44-53). */
typedef struct ppg_config {
    int width, height;          /* must be multiples of 8 (PPGExtractor.cpp:163) */
    float K[9];                 /* row-major 3x3 intrinsics */
    float D[4];                 /* distortion; parity is pinned for D = 0 only */
    int fisheye;
    int device_id;
    const char* weights_path;   /* flat fp32 blob written by tools/export_weights.py */
    int max_batch;              /* frames per ppg_extract_batch call (>= 1) */
    int desc_dim;               /* DESC_DIM_SIZE, 256 */
    float junction_thresh;      /* JUNCTION_THRESH */
    int junction_nms_radius;    /* JUNCTION_NMS_RADIUS (only 4 is built) */
    unsigned junction_max_num;  /* JUNCTION_MAX_NUM */
    float line_valid_thresh;    /* LINE_VALID_THRESH */
    float line_valid_ratio;     /* LINE_VALID_RATIO */
    float line_dist_thresh;     /* LINE_DISTTHRESH */
    int heatmap_refine_sz;      /* HEATMAP_REFINE_SZ */
    float line_heatmap_thresh;  /* LINE_HEATMAP_THRESH */
    float line_inlier_rate;     /* LINE_INLIER_RATE */
    int precision;              /* 0: fp32 operands, fp32 MFMA accumulate (parity path: integer outputs exact);
                                   1: the convolutions (all but the one-channel first layer and the final 4->2 one) on the
                                      bf16 MFMA with every operand split
                                      into bf16 hi + lo and three products (hi*hi + hi*lo + lo*hi), fp32 accumulate:
                                      descriptors within 1e-3 of the fp32 path, key-point sets equal up to near-ties;
                                   2: the same layers with plain bf16 operands (outside the parity bar; for the
                                      deviation report only);
                                   3: as 1 with fp16 halves (operands pre-scaled by exact powers of two so that both halves
                                      stay in fp16's range): about 22 significant bits per operand, results within twice
                                      the rounding noise of an fp32 evaluation.  Range: activations entering a 3x3 layer
                                      saturate at +-2047 and |weight| at 63.9 (the shipped networks reach 212 and 1.55
                                      on full-contrast images).
                                   Descriptor sampling, softmax, NMS, matching: fp32 always */
} ppg_config;

void ppg_default_config(ppg_config* cfg, int width, int height);

/* PPGExtractor::PPGExtractor (PPGExtractor.cpp:55-107): loads the weights, allocates every device
 * and pinned buffer, warms up.  Returns NULL on failure; ppg_last_error(NULL) then holds the reason. */
ppg_ctx* ppg_create(const ppg_config* cfg);
void ppg_destroy(ppg_ctx* ctx);
const char* ppg_last_error(const ppg_ctx* ctx);

/* A context for matching only.  The reference builds a Matcher on the stack wherever it needs one, in three threads at
 * once (tracking: system/src/Tracking.cpp:524, 761, 803, 992; local mapping: system/src/LocalMapping.cpp:214; loop
 * closing: system/src/LoopClosing.cpp:408-409, 786, 1061); a context is not thread-safe, so each of those threads owns
 * one.  This kind loads no weights, allocates no network or frame buffers and does no warm-up: it costs five streams and
 * a few bytes until its first call grows the matcher's work arrays.  Of cfg it reads width, height, K, D, fisheye (the
 * image bounds of the frame grid), device_id and junction_max_num.  Every matching/ entry point, the grid, the
 * vocabulary transform of host descriptors and ppg_match_nn_noalloc work on it; the extractor's entry points and the
 * ones that use resident frames (ppg_match_frames, ppg_match_prev_batch, ppg_bow_transform_batch) return PPG_EINVAL. */
ppg_ctx* ppg_create_matcher(const ppg_config* cfg);

/* The ten tunables are public mutable statics that the reference reads on every frame (feature/include/PPGExtractor.h:
 * 105-114).  This call applies the ones that do not size a buffer — junction_thresh, line_valid_thresh,
 * line_valid_ratio, line_dist_thresh, heatmap_refine_sz, line_heatmap_thresh, line_inlier_rate — from cfg to the next
 * frames, without touching weights or buffers.  A cfg whose desc_dim, junction_nms_radius or junction_max_num differs
 * from the context's returns PPG_ECAP (those need a new context); pending batches must have been waited for. */
int ppg_set_tunables(ppg_ctx* ctx, const ppg_config* cfg);

/* Results of PPGExtractor::run for one frame (PPGExtractor.cpp:118-147) as structure-of-arrays.
 * Buffers are owned by the context and stay valid until the next extract call on it.
 *   kp_xy        mvKeyPoints[i].mPos as integers (x, y)           (PPGExtractor.cpp:194)
 *   kp_un        mPosUn (u, v)                                    (:232)
 *   kp_out       mbOut                                            (:230-231)
 *   desc         normDesc rows, n x desc_dim                      (:133-139)
 *   edge_idx     mvKeyEdges[e].{startIdx,endIdx}; edge_score = lscore (:433-441)
 *   conn_off/idx CSR of KeyPointEx::mvConnected (edge ids, reference order)
 *   coline_*     CSR of KeyPointEx::mvColine (pairs of point ids)  (:427) */
typedef struct ppg_frame_out {
    int n;
    const int32_t* kp_xy;
    const float* kp_score;
    const float* kp_un;
    const uint8_t* kp_out;
    const float* desc;
    int n_edges;
    const uint32_t* edge_idx;
    const float* edge_score;
    const int32_t* conn_off;
    const uint32_t* conn_idx;
    const int32_t* coline_off;
    const uint32_t* coline_pairs;
} ppg_frame_out;

/* PPGExtractor::run (feature/include/PPGExtractor.h:50-51) for one host image (8-bit, 1 channel). */
int ppg_extract(ppg_ctx* ctx, const uint8_t* img, int row_stride, ppg_frame_out* out);
/* The same for n <= max_batch independent frames in one pass (multi-camera / multi-sequence mode;
 * no counterpart in the reference, whose batch size is fixed to 1 at PPGExtractor.cpp:151). */
int ppg_extract_batch(ppg_ctx* ctx, const uint8_t* const* imgs, int n, int row_stride, ppg_frame_out* outs);
/* As ppg_extract_batch with the n images already resident in device memory, contiguous n*H*W bytes. */
int ppg_extract_batch_device(ppg_ctx* ctx, const uint8_t* d_imgs, int n, ppg_frame_out* outs);

/* Pipelined form for throughput: ppg_submit_batch_device enqueues the whole extract of n device-resident
 * frames (and, with match_prev != 0, the all-pairs SearchForInitialization match of every frame b against
 * frame b of the previously submitted batch) plus the copies to pinned host memory, and returns without
 * waiting.  ppg_wait_batch blocks until the OLDEST submitted batch is on the host, finishes its line
 * graphs and fills outs (and matches12: n rows of junction_max_num entries; n_matches: n counts; both
 * optional).  It returns the batch's frame count (> 0) or a negative error.  At most four batches may be
 * in flight.  The buffers behind `outs` belong to a ring of five slots: they survive the first ppg_submit_batch*
 * after this wait and are overwritten by the second one (with fewer batches in flight they live longer; never rely on
 * more than that one submit).
 * No counterpart in the reference (PPGExtractor::run blocks, PPGExtractor.cpp:125). */
int ppg_submit_batch_device(ppg_ctx* ctx, const uint8_t* d_imgs, int n, int match_prev, float nnratio);
/* The same for host images: they are copied to pinned staging memory and uploaded before the call returns (the upload
 * runs beside the convolutions of the batches submitted before). */
int ppg_submit_batch(ppg_ctx* ctx, const uint8_t* const* imgs, int n, int row_stride, int match_prev, float nnratio);
int ppg_wait_batch(ppg_ctx* ctx, ppg_frame_out* outs, int32_t* matches12, int* n_matches);

/* ---- the public pipeline stages of PPGExtractor (feature/include/PPGExtractor.h:56-65), exposed so
 * that each can be checked on its own.  They operate on the context's resident tensors for the
 * first n frames. ---- */
int ppg_inference(ppg_ctx* ctx, const uint8_t* imgs, int n, int on_device);  /* :149-156 (K0-K8)        */
int ppg_detect_keypoints(ppg_ctx* ctx, int n);                               /* :158-234 (K9 + NMS)     */
int ppg_detect_lines(ppg_ctx* ctx, int n);                                   /* :236-442 (K10 + graph)  */
int ppg_gen_descriptors(ppg_ctx* ctx, int n);                                /* :515-538 (K11 + K12)    */
/* Fill `outs` from the stage results (the copy-out at :130-146). */
int ppg_collect(ppg_ctx* ctx, int n, ppg_frame_out* outs);

/* Resident tensors that can be read back or overwritten for stage-wise checks.  Layouts are the
 * device layouts: channels-last (NHWC) fp32, one frame at a time. */
enum ppg_tensor {
    PPG_T_FEAT = 0,      /* featureMap  [h][w][128]          (PPGExtractor.cpp:152) */
    PPG_T_JUNC = 1,      /* junctions   [h][w][65]           (:153) */
    PPG_T_HEAT_RAW = 2,  /* heatmap     [H][W][2]            (:154) */
    PPG_T_DESCMAP = 3,   /* descriptors [h][w][256]          (:155) */
    PPG_T_SCORE = 4,     /* junc_pred   [H][W]               (:162) */
    PPG_T_HEAT = 5,      /* heatmap_score [H][W] before refineHeatMap (:242) */
    PPG_T_HEAT_REFINED = 6, /* after refineHeatMap            (:243-256) */
    PPG_T_HEAT_REMAPPED = 7 /* after cv::remap, exists only when D[0] != 0 (:261-262); read-only */
};
size_t ppg_tensor_elems(const ppg_ctx* ctx, int which);
int ppg_read_tensor(ppg_ctx* ctx, int which, int frame, float* dst, size_t n_elems);
int ppg_write_tensor(ppg_ctx* ctx, int which, int frame, const float* src, size_t n_elems);

/* ---- matching/ ---- */

/* DescriptorDistance (feature/include/MapPoint.h:25, feature/src/MapPoint.cpp:22-29) for every pair:
 * D[i*nB + j] = || A[i] - B[j] ||_2, rows of `dim` floats.  Host pointers.  The sum runs over
 * d = 0..dim-1 as one fused-multiply-add chain per pair, then a correctly rounded square root. */
int ppg_dist_matrix(ppg_ctx* ctx, const float* A, int nA, const float* B, int nB, int dim, float* D);

/* DescriptorDistance for ONE pair on the host (feature/src/MapPoint.cpp:22-29; also called from MapPoint.cpp:277): the
 * same chain as every kernel of the library, bit for bit, without a launch or a context — what the free function
 * `float DescriptorDistance(const cv::Mat&, const cv::Mat&)` becomes. */
float ppg_descriptor_distance(const float* a, const float* b, int dim);

/* DescriptorDistance for listed pairs only: row i of A against the rows cand_idx[cand_off[i] .. cand_off[i+1]-1] of B,
 * D[k] for the k-th listed pair.  The inner arithmetic of every candidate-list matcher whose control flow stays on the
 * host: Matcher::SearchByBoW (matching/src/Matcher.cpp:393-477, 663-754) and SearchForTriangulation (:767-885), where
 * the lists are the shared DBoW3 FeatureVector nodes. */
int ppg_dist_lists(ppg_ctx* ctx, const float* A, int nA, const float* B, int nB, int dim,
                   const int32_t* cand_off, const int32_t* cand_idx, float* D);

#define PPG_MATCH_PLAIN 0  /* rows independent: first argmin, best and second-best distance only      */
#define PPG_MATCH_INIT 1   /* Matcher::SearchForInitialization claim semantics (Matcher.cpp:591-644)  */

/* Nearest-neighbour search of every row of A among candidate rows of B.
 * cand_off/cand_idx: CSR candidate lists per A row in the order the reference would visit them
 * (Frame::GetFeaturesInArea order, map/src/Frame.cpp:294-311); NULL = every B row, ascending.
 * skipB (optional, nB bytes): rows of B never considered (e.g. slots already holding an observed
 * map point, Matcher.cpp:67-69).
 * PLAIN: match12[i] = first argmin or -1, best[i], second[i] (1e6 when absent, Matcher.cpp:62).
 * INIT : sequential over i with the reference's claim rule; match12[i] set when
 *        best <= th && best < second*ratio.  Returns the number of matches through *n_matches. */
int ppg_match_nn(ppg_ctx* ctx, const float* A, int nA, const float* B, int nB, int dim,
                 const int32_t* cand_off, const int32_t* cand_idx, const uint8_t* skipB,
                 float th, float ratio, int mode,
                 int32_t* match12, float* best, float* second, int* n_matches);

/* ppg_match_nn for callers that cannot allocate.  No counterpart in the reference: it exists for code that must not
 * call hipMalloc/hipFree — a stream under HIP graph capture, a tight serving loop, a harness that forbids allocation.
 * It is ppg_match_nn over every B row (no candidate lists, no skipB) with inputs AND outputs in device memory of the
 * context's GPU: dA [nA][dim], dB [nB][dim], d_match12 [nA], d_best [nA], d_second [nA], d_n_matches [1] (may be NULL
 * in PLAIN mode).  Its work memory (the nA x nB distance matrix and the eight-nearest lists, 4.06 MiB) is a fixed
 * device array inside the library that every GPU holds until the process ends, so nA and nB are limited to
 * PPG_NOALLOC_MAX_ROWS (PPG_ECAP beyond that; ppg_match_nn has no such limit).  The call allocates and frees nothing.
 * stream: a hipStream_t (as void*) to enqueue on, without waiting — or NULL: the context's own stream, and the call
 * returns when the results are in place.  All calls on one GPU share the one work array, and the library orders them:
 * host threads take a per-GPU lock for the duration of the call (a call on the context's own stream holds it until its
 * results are in place), and a call on another stream than the one before it first makes its stream wait for an event
 * recorded behind that call, so two contexts, streams or threads never use the array at once.  The one exception is a
 * stream under graph capture: no event can tie a captured call to work outside its graph, so launches of such a graph
 * must not overlap other calls on that GPU. */
#define PPG_NOALLOC_MAX_ROWS 1024
int ppg_match_nn_noalloc(ppg_ctx* ctx, const float* dA, int nA, const float* dB, int nB, int dim,
                         float th, float ratio, int mode,
                         int32_t* d_match12, float* d_best, float* d_second, int* d_n_matches, void* stream);

/* Mutual nearest neighbours, the agreement check of Matcher::SearchBySim3 (matching/src/Matcher.cpp:1316-1332): row i of
 * A takes its first argmin j among its candidates when that distance is <= th (vnMatch1, :1230-1260), row j of B takes its
 * first argmin among ITS candidates likewise (vnMatch2, :1287-1312), and match12[i] = j only if both agree.  cand12 /
 * cand21: CSR candidate lists for the two directions (the windows around the two projections); both NULL = all pairs.
 * Both searches run on the GPU (match_plain_kernel, wave-shuffle argmin); *n_matches = nFound. */
int ppg_match_mutual(ppg_ctx* ctx, const float* A, int nA, const float* B, int nB, int dim,
                     const int32_t* cand12_off, const int32_t* cand12_idx,
                     const int32_t* cand21_off, const int32_t* cand21_idx,
                     float th, int32_t* match12, int* n_matches);

/* Matcher::SearchForInitialization (matching/include/Matcher.h:40, Matcher.cpp:582-651).
 * pos2: F2.mvKeysUn[i].mPos; prev_matched (2*n1, in/out) = vbPrevMatched; window <= 0 = all pairs. */
int ppg_search_for_initialization(ppg_ctx* ctx, float nnratio,
                                  const float* desc1, int n1,
                                  const float* desc2, int n2, const float* pos2,
                                  float* prev_matched, int window, int32_t* matches12, int* n_matches);

/* All-pairs SearchForInitialization between two frames of the last ppg_extract_batch call whose
 * descriptors are still resident on the GPU (frame_a = F1, frame_b = F2); only the N1 match
 * indices come back to the host. */
int ppg_match_frames(ppg_ctx* ctx, int frame_a, int frame_b, float nnratio, int32_t* matches12, int* n_matches);

/* The same for every stream of a batch: frame b of the PREVIOUS extract call (F1) against frame b of
 * the LAST one (F2), b < n, all resident on the GPU.  matches12 holds n rows of junction_max_num
 * entries (-1 beyond a frame's key-point count); n_matches (optional) holds n counts.  No counterpart in
 * the reference, which matches one pair per call. */
int ppg_match_prev_batch(ppg_ctx* ctx, int n, float nnratio, int32_t* matches12, int* n_matches);

/* Frame::GetFeaturesInArea (map/src/Frame.cpp:262-315) over the 64x48 grid built as
 * Frame::AssignFeaturesToGrid / PosInGrid do (:138-156, :317-327) with the image bounds of
 * GeometricCamera::InitializeImageBounds (sensors/src/GeometricCamera.cpp:26-61).  Host-side integer
 * logic (the candidate generator of every windowed matcher). */
typedef struct ppg_grid ppg_grid;
ppg_grid* ppg_grid_build(const ppg_config* cfg, const float* pos, int n);
void ppg_grid_free(ppg_grid* g);
int ppg_features_in_area(const ppg_grid* g, float x, float y, float r, int32_t* out, int cap);

/* GeometricCamera::IsInImage (sensors/src/GeometricCamera.cpp:21-24) with the bounds the grid was built with. */
int ppg_grid_is_in_image(const ppg_grid* g, float x, float y);

/* Frame::CheckInFrustum (map/src/Frame.cpp:223-260) for m map points at once (host arithmetic: it prepares the
 * mTrackProjX/Y, mTrackViewCos, mbTrackInView inputs of ppg_extend_map_matches).  pose = camera-from-world rotation
 * (row-major 3x3) followed by the translation, 12 floats; "R * P + t", ".norm()" and ".dot()" are evaluated left to
 * right in float.  Outputs per point: in_view (0/1), proj (2 floats, -1 when not in view), depth and view_cos
 * (-1 / unchanged semantics of the reference: depth -1 when not in view; view_cos written only when in view). */
int ppg_check_in_frustum(const ppg_grid* bounds, const float* K9, const float* pose12, int m,
                         const float* pos, const float* normal, const float* min_dist, const float* max_dist,
                         float viewing_cos_limit, uint8_t* in_view, float* proj, float* depth, float* view_cos);

/* The windowed projection matchers: Matcher::SearchByProjection(Frame&, const Frame&, th) (matching/src/Matcher.cpp:31-87),
 * SearchByProjection(Frame&, vector<MapPoint*>&, th) (:89-189), SearchByProjection(KeyFrame*, Sim3f&, ...) (:479-568),
 * Fuse(KeyFrame*, vector<MapPoint*>&, th) (:897-1036) and Fuse(KeyFrame*, Sim3f&, ...) (:1038-1135) share one inner
 * loop: per projected map point, in order, the candidates of GetFeaturesInArea(uv, radius) in grid order, some slots
 * skipped, first argmin of DescriptorDistance, an accept rule, and (for two of them) an accepted match blocks its
 * slot for the rows after it.  The caller keeps the geometry gates (pose, depth, scale range, viewing angle) and
 * the map side effects; this call evaluates the loop with every candidate distance computed on the GPU.
 *   row_ok[i] == 0        the row is skipped (a gate failed before GetFeaturesInArea)
 *   slot_blocked (in/out) slots never taken: F.mvpMapPoints[i2]->Observations() > 0 (:67-69, :154-156) or
 *                         vpMatched[idx] != nullptr (:546); may be NULL
 *   row_claims[i] != 0    an accepted match of row i sets slot_blocked[best] (pMP->Observations() > 0 for :80,
 *                         always for :562); may be NULL (Fuse)
 *   e2_max >= 0           candidates whose squared distance to uv exceeds it are skipped (:992-997: 5.99)
 *   accept                best <= th, and when ratio > 0 also !(best > ratio * second) (:177-180)
 * best_idx[i] = accepted slot or -1, best_dist[i] = its distance (1e6 when the row had no candidate). */
typedef struct ppg_proj_args {
    const ppg_grid* grid;
    const float* desc2;
    int n2;
    const float* desc1;
    int n1;
    const float* uv;
    const float* radius;
    const uint8_t* row_ok;
    const uint8_t* row_claims;
    uint8_t* slot_blocked;
    float th;
    float ratio;
    float e2_max;
} ppg_proj_args;
int ppg_match_projected(ppg_ctx* ctx, const ppg_proj_args* args, int32_t* best_idx, float* best_dist, int* n_matches);

/* MapPoint::ComputeDistinctiveDescriptors (feature/src/MapPoint.cpp:234-302) for P map points at once.  desc holds
 * the observed descriptors of all points back to back (the rows the reference pushes into vDescriptors, :253-263, in
 * map order); point p owns rows seg_off[p] .. seg_off[p+1]-1.  All pair distances of a point come from one GPU
 * launch; per row the median vDists[0.5*(N-1)] of the sorted row (:288-290) and the first row whose median is below
 * the running best (initially 1.0, :284-295) are taken on the host.  best[p] = row index inside the point's segment
 * (0 for an empty or single-row segment, as BestIdx starts at 0). */
int ppg_distinctive_descriptors(ppg_ctx* ctx, const float* desc, const int32_t* seg_off, int n_points, int32_t* best);

/* Matcher::ExtendMapMatches (matching/include/Matcher.h:54, Matcher.cpp:203-381) on flattened state.
 * Map points are numbered 0..m-1 in vpMapPoints order.  Per point: descriptor, mTrackProjX/Y,
 * mTrackViewCos, mbTrackInView, isBad(), Observations(), and getEdges() as CSR entries carrying the
 * map-edge id, the id of theOtherPt() (-1 for nullptr) and the flag !isBad() && mbValid. */
typedef struct ppg_mapgraph {
    int m;
    const float* desc;
    const float* proj;
    const float* viewcos;
    const uint8_t* in_view;
    const uint8_t* bad;
    const int32_t* nobs;
    const int32_t* edge_off;
    const int32_t* edge_id;
    const int32_t* edge_other;
    const uint8_t* edge_ok;
} ppg_mapgraph;
/* tracked[m]: mnTrackedbyFrame == F.mnId (in/out); frame_mp[n]: F.mvpMapPoints as ids or -1 (in/out);
 * frame_me[n_edges]: F.mvpMapEdges as ids or -1 (in/out). *n_matches = the reference's return value. */
int ppg_extend_map_matches(ppg_ctx* ctx, float nnratio, float th, const ppg_mapgraph* g,
                           const float* desc, const float* pos_un, int n,
                           const uint32_t* edge_idx, int n_edges,
                           const int32_t* conn_off, const uint32_t* conn_idx,
                           uint8_t* tracked, int32_t* frame_mp, int32_t* frame_me, int* n_matches);

/* HIP stream the context launches on (as a void* hipStream_t). */
void* ppg_stream(ppg_ctx* ctx);

/* Device-time accounting with HIP events recorded on that stream around every launch (no counterpart in
 * the reference, which has no timers).  ppg_profile(ctx, 1) resets and enables, (ctx, 0) disables;
 * (ctx, 2) resets and times only PPG_PROF_CONV1A on the device (two events per batch instead of two per
 * kernel, which perturbs a throughput measurement far less) plus the host-side classes;
 * ppg_profile_read returns accumulated milliseconds and the number of timed launches per class. */
#define PPG_PROF_CONV3X3 0     /* conv_mfma_kernel<3,2,8,false>: the 3x3 convolutions with >= 64 couts after conv1b */
#define PPG_PROF_CONV_OTHER 1  /* the 1x1 convolutions and the 16-cout 3x3 one */
#define PPG_PROF_CONV1A 2      /* conv_mfma_kernel<3,2,8,true>: conv1a fused into conv1b (or conv1a_kernel alone) */
#define PPG_PROF_POST 3        /* detectKeyPoint: score map, NMS, ranking, undistortion */
#define PPG_PROF_MATCH 4
#define PPG_PROF_HOST_WAIT 5   /* host wall time blocked in hipStreamSynchronize inside ppg_collect */
#define PPG_PROF_HOST_TAIL 6   /* host wall time of the order-dependent line-graph tail */
#define PPG_PROF_HOST_SUBMIT 7 /* host wall time inside ppg_submit_batch* (launch + copy enqueue) */
#define PPG_PROF_HOST_FINISH 8 /* host wall time inside ppg_wait_batch */
#define PPG_PROF_HOST_TAIL_FRAMES 9 /* sum over frames of the per-frame tail time (launches = frames) */
#define PPG_PROF_LINES 10      /* detectLines on the device: heat tail, refine, remap, pair probe, scores */
#define PPG_PROF_DESC 11       /* genPointDescriptor */
#define PPG_PROF_N 12
int ppg_profile(ppg_ctx* ctx, int enable);
int ppg_profile_read(ppg_ctx* ctx, double* ms, long long* launches);
/* Algorithmic FLOPs (2 x MACs of the layers' shapes) enqueued per class since ppg_profile(ctx, 1); conv classes only. */
int ppg_profile_flops(ppg_ctx* ctx, double* flops);

/* ---- DBoW3 vocabulary (map/src/Frame.cpp:131, :338: pMap->mpVoc->transform(vCurrentDesc, mBowVec, mFeatVec, 4)) ----
 * DBoW3 is a third-party dependency of the reference that is not vendored; its binary vocabulary format and its tree
 * descent are restated from DBoW3's published source. */
typedef struct ppg_vocab ppg_vocab;
/* DBoW3::Vocabulary::load of a binary vocabulary (Vocabulary/voc_euroc_9x3.gz and voc_tum_9x3.gz are such files despite
 * the suffix; QuickLZ level-1 packets are decoded).  Host only.  NULL on failure, reason in ppg_last_error(NULL). */
ppg_vocab* ppg_vocab_load(const char* path);
void ppg_vocab_free(ppg_vocab* v);
/* k, L, node count (root included), word count, descriptor length, DBoW3::ScoringType, DBoW3::WeightingType; any pointer may be NULL */
int ppg_vocab_info(const ppg_vocab* v, int* k, int* L, int* n_nodes, int* n_words, int* dim, int* scoring, int* weighting);
/* one node of the tree: parent (-1 for the root), weight, word id (-1 for inner nodes), children in visiting order, descriptor */
int ppg_vocab_node(const ppg_vocab* v, int node, int* parent, double* weight, int* word_id, int* n_children,
                   const int32_t** children, const float** desc);
/* Vocabulary::transform(feature, word_id, weight, nid, levelsup) for n descriptors (host pointers, rows of the
 * vocabulary's descriptor length) in one launch: the leaf's word id and weight and the id of the node passed at level
 * L - levelsup (0, the root, when that is <= 0 — the reference's levelsup = 4 with its L = 3 vocabularies). */
int ppg_bow_transform(ppg_ctx* ctx, const ppg_vocab* v, const float* desc, int n, int levelsup,
                      int32_t* word_id, double* weight, int32_t* node_id);
/* The same for the first n_frames frames of the last ppg_extract* call, whose descriptors are still on the GPU: outputs
 * hold n_frames rows of junction_max_num entries (-1 / 0 beyond a frame's key points). */
int ppg_bow_transform_batch(ppg_ctx* ctx, const ppg_vocab* v, int n_frames, int levelsup,
                            int32_t* word_id, double* weight, int32_t* node_id);
/* The two containers transform() fills, from the per-feature results (host): BowVector as (word id, value) ascending by
 * word — weights added per word (TF_IDF, TF) or first kept (IDF, BINARY), then normalised as the vocabulary's scoring
 * type demands — and FeatureVector as CSR: node ids ascending, the feature indices of each in feature order.  Features
 * whose word weight is 0 enter neither.  Capacities: n entries each (fv_off n + 1). */
int ppg_bow_vectors(const ppg_vocab* v, const int32_t* word_id, const double* weight, const int32_t* node_id, int n,
                    int32_t* bow_word, double* bow_value, int* n_bow,
                    int32_t* fv_node, int32_t* fv_off, int32_t* fv_idx, int* n_fv);

/* Where the order-dependent tail of detectLines (feature/src/PPGExtractor.cpp:315-441) ran since ppg_create: *frames =
 * frames collected, *host_frames = those whose tail the host statement had to redo (the device tail hands a frame over
 * when a double-precision atan2 / sin result sits within 64 ulps of a float rounding boundary, where another libm could
 * round the other way, or when a frame exceeds its work arrays; PPG_LINES_HOST=1 sends every frame there).  No
 * counterpart in the reference. */
int ppg_tail_stats(ppg_ctx* ctx, long long* frames, long long* host_frames);

/* ---- multi-GPU mode: independent camera streams sharded over the GPUs of one node, one process (or thread) per GPU ----
 * The reference is one process bound to one device (feature/src/PPGExtractor.cpp:41: torch::Device(torch::kCUDA, 0);
 * one extractor, system/src/Tracking.cpp:51); these entry points are what replaces that hard-coded device when several
 * cameras are extracted at once.  Streams never exchange activations.  The only collective is the global-shutter
 * barrier: an all-reduce(MIN) over RCCL of each rank's newest finished frame group (4 bytes, latency-bound), so that
 * the results of group k are released on every GPU together.  RCCL is loaded at the first of these calls (an RCCL image
 * the process already maps is used; else PPG_RCCL_LIB; else librccl.so.1 of the ROCm installation).
 *
 *   rank 0:      ppg_shutter_unique_id(id); hand the 128 bytes to every rank (file, socket, launcher, ...)
 *   every rank:  sh = ppg_shutter_create(ctx, rank, world, id)        collective: returns when all ranks have joined
 *   per step:    ppg_wait_batch(...); ppg_shutter_release(sh, k, &released)   never waits for the collective
 *   at the end:  ppg_shutter_drain(sh, &released); ppg_shutter_destroy(sh)
 *
 * ppg_shutter_release starts the all-reduce of `group` on a stream of the shutter's own (the result lands in pinned
 * memory behind an event) and reports in *released the newest group known to be finished everywhere, which may lag by
 * the barriers still in flight (at most eight; the ninth call waits for the oldest).  Every rank must call it the same
 * number of times.  ppg_shutter_poll reads what has arrived without starting anything; ppg_shutter_drain waits for
 * every barrier started.  A shutter belongs to the thread that drives its context. */
typedef struct ppg_shutter ppg_shutter;
#define PPG_SHUTTER_ID_BYTES 128
int ppg_shutter_unique_id(uint8_t* id /* PPG_SHUTTER_ID_BYTES */);
/* NULL on failure, reason in ppg_last_error(NULL).  ctx supplies the GPU and junction_max_num. */
ppg_shutter* ppg_shutter_create(ppg_ctx* ctx, int rank, int world, const uint8_t* id);
void ppg_shutter_destroy(ppg_shutter* sh);
int ppg_shutter_release(ppg_shutter* sh, int group, int* released);
int ppg_shutter_poll(ppg_shutter* sh, int* released);
int ppg_shutter_drain(ppg_shutter* sh, int* released);
/* barriers started so far and those whose result has not been read yet; either pointer may be NULL */
int ppg_shutter_stats(const ppg_shutter* sh, long long* started, int* pending);
/* A stereo rig split over two GPUs: swaps a frame's descriptors with rank `partner` (ncclSend + ncclRecv in one group,
 * one fixed-size message of junction_max_num x 256 floats + the row count each way: 1 MiB at 1024 key points, a few
 * microseconds on one xGMI link), after which each side matches left against right with ppg_match_nn / ppg_match_mutual.
 * desc: n rows of 256 floats (host); partner_desc: room for junction_max_num rows (host); *n_partner = rows received.
 * Both sides call it for every frame, with contexts of the same junction_max_num.  Blocks until the swap is done. */
int ppg_stereo_swap(ppg_shutter* sh, int partner, const float* desc, int n, float* partner_desc, int* n_partner);
/* The same swap and the left-right match in one call, for a frame whose descriptors are still on the GPU: frame `frame` of the
 * batch that the last ppg_extract* / ppg_wait_batch on the shutter's context handed out (the ring of result slots keeps it
 * untouched until the next of those calls, whatever is submitted in between).  The rows go GPU to GPU (nothing of them crosses PCIe), and this side's rows are
 * matched against the partner's as Matcher::SearchForInitialization does with a window that holds every feature
 * (matching/src/Matcher.cpp:582-651: all pairs, TH_LOW, `nnratio`), on a matcher context and stream of the shutter's own,
 * beside the extraction pipeline.  matches12: junction_max_num entries (-1 beyond this frame's key points);
 * n_matches, n_partner may be NULL.  Both sides call it for every frame.  Blocks until the result is in place.
 * The reference has no counterpart (one process, one device; its stereo matching happens on the host). */
int ppg_stereo_match(ppg_shutter* sh, int partner, int frame, float nnratio, int32_t* matches12, int* n_matches, int* n_partner);
/* The two halves of ppg_stereo_match for a loop that must not wait: _start enqueues swap and match and returns, _finish
 * waits for them and hands out the result (typically one frame later).  One match at a time per shutter; ppg_stereo_swap
 * shares its buffers and is refused in between. */
int ppg_stereo_match_start(ppg_shutter* sh, int partner, int frame, float nnratio);
int ppg_stereo_match_finish(ppg_shutter* sh, int32_t* matches12, int* n_matches, int* n_partner);

#ifdef __cplusplus
}
#endif
#endif /* PPGFRONT_H */
