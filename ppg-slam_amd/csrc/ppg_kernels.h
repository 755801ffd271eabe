// Launch interface between the host pipeline (ppg_front.cpp) and the gfx950 kernels (conv_kernels.hip, conv_ws.hip, post_kernels.hip, match_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ppg {

// ---- implicit-GEMM convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32) ----
enum ConvMode { CONV_PLAIN = 0, CONV_POOL2 = 1, CONV_PS2 = 2 };

// One layer served by a launch: its cout blocks start at nblk0; bias/alpha/beta are indexed by the
// layer's own packed cout.
struct ConvGroup {
    float* out = nullptr;
    const float* bias = nullptr;
    const float* alpha = nullptr;  // eval-mode batch-norm scale or nullptr
    const float* beta = nullptr;   // eval-mode batch-norm shift
    int nblk0 = 0;
    int Cout = 0;
    int mode = 0;                  // ConvMode
    int relu = 0;
};

struct ConvArgs {
    const float* in;     // [B][H][W][Cin] channels-last (unused when img != nullptr)
    const uint8_t* img = nullptr;  // fused first layer: [B][H][W] image; this conv's input is conv1a(img)
    const float* w1a = nullptr;    // [9][64] conv1a weights, tap-major
    const float* b1a = nullptr;    // [64]
    const float* wpk;    // packed per (cout block, cin chunk), the blocks of all groups concatenated
                         // (prec != 0: the bf16 hi/lo image of conv_bf16_kernel, same size)
    int* sat = nullptr;  // prec 3: set to 1 when an activation saturates the fp16 split
    int prec = 0;        // 0: fp32 MFMA; 1: bf16 hi/lo split, three products; 2: plain bf16; 3: fp16 hi/lo split (3x3, nt = 2 only)
    int B, H, W, Cin;
    int tiles_x, tiles_y, nblk;  // nblk = cout blocks of all groups together
    int persist = 0;             // fp32 kernels: > 0 caps the work-groups per cout block (each then takes several tiles)
    int* tile_counter = nullptr; // 2 * nblk zeroed ints the tiles are handed out through when persist applies (the kernel zeroes them again)
    int mw = 32;         // tile shape: 32 -> 8 x 32 pixels, 16 -> 16 x 16 pixels (3x3, 64-cout blocks only)
    // XCD-aware launch order.  Work-groups of a launch go round the eight XCDs (work-group L runs on XCD L % 8), each
    // with its own L2.  xcd_group = G > 0 (the launch function lowers it to a divisor of the cout blocks) turns a launch
    // of one work-group per tile into a one-dimensional one of 8 * xcd_tiles8 * (cout blocks) work-groups in which XCD x
    // serves the band of tiles x * xcd_tiles8 ... in order (xcd_band; otherwise the tiles t with t % 8 == x) and the G
    // cout blocks of a group that share a tile are dispatched back to back onto that XCD: neighbouring tiles find their
    // common halo, and the blocks of a group their whole input tile, in the XCD's L2 instead of fetching it again.
    // Work-group L: x = L % 8, k = L / 8, block in group k % G, tile slot (k / G) % xcd_tiles8, group k / G / xcd_tiles8.
    int xcd_group = 0;
    int xcd_band = 1;
    int xcd_tiles8 = 0;  // set by the launch function: ceil(tiles / 8)
    int ws = 0;          // 1: the wave-specialised kernel (conv_ws.hip; reduced-precision 3x3 layers with 64-cout blocks, Cin >= 64)
    int quarter = 0;     // 1: quarter-size work-groups (fp32 3x3, 64-cout blocks): 4 x 32 (mw 32) or 8 x 16 (mw 16) pixels x 32 couts
    int ngroups = 1;
    ConvGroup g[3];
};

// ks: 1 or 3; nt: 32-wide cout tiles per wave (1 or 2)
hipError_t launch_conv(const ConvArgs& a, int ks, int nt, hipStream_t s);
// the wave-specialised form of a reduced-precision 3x3 launch (a.prec != 0, 64-cout blocks, Cin a multiple of 16 and >= 64);
// a.tile_counter must point at kTileCounters zeroed ints
hipError_t launch_conv_ws(const ConvArgs& a, hipStream_t s);
// cin chunk depth the kernel for (ks) streams through LDS; packers must use the same value
int conv_chunk(int ks);

struct Conv1aArgs {
    const uint8_t* img;  // [B][H][W]
    const float* w;      // [9][64]
    const float* bias;   // [64]
    float* out;          // [B][H][W][64]
    int B, H, W;
};
hipError_t launch_conv1a(const Conv1aArgs& a, hipStream_t s);

// Copies `bytes` (a multiple of 16, both pointers 16-byte aligned) from pinned host memory to device memory with a kernel.
hipError_t launch_upload(const void* host_src, void* dst, size_t bytes, hipStream_t s);

// ---- junction score map + non-maximum suppression ----
struct NmsArgs {
    const float* junc;    // [B][h][w][65]
    float* score;         // [B][H][W]
    uint8_t* state;       // [B][H*W]  0 none, 1 undecided, 2 kept, 3 suppressed
    uint32_t* resid;      // [B][resid_cap] undecided pixels after the two parallel passes
    uint32_t* kept;       // [B][kept_cap]
    unsigned long long* kept_key;  // [B][kept_cap]
    int* counters;        // [B][8]: 0 resid, 1 kept, 2 line candidates, 3 n keypoints
    int resid_cap, kept_cap;
    int B, H, W;
    float thresh;
    int max_num;
    // camera (undistortion of the kept points)
    float fx, fy, cx, cy, d0, d1, d2, d3;
    int fisheye;
    // outputs, per frame stride max_num
    int32_t* kp_xy;       // [B][max_num][2]
    float* kp_score;      // [B][max_num]
    float* kp_un;         // [B][max_num][2]
    uint8_t* kp_out;      // [B][max_num]
};
hipError_t launch_score_map(const NmsArgs& a, hipStream_t s);
hipError_t launch_nms(const NmsArgs& a, hipStream_t s);

// ---- line heat map ----
struct HeatArgs {
    const float* ps;      // [B][H][W][4] output of the last pixel-shuffle block
    const float* w;       // [2][4] 1x1 conv
    const float* bias;    // [2]
    float* raw;           // [B][H][W][2] (kept for checks) or nullptr
    float* prob;          // [B][H][W]
    float* refined;       // [B][H][W]
    int B, H, W;
    int refine_sz;
    float valid_thresh, valid_ratio;
};
// cv::remap(heat, heat, mX, mY, INTER_LINEAR) of feature/src/PPGExtractor.cpp:261-262 (BORDER_CONSTANT 0)
struct RemapArgs {
    const float* src;   // [B][H][W]
    float* dst;         // [B][H][W]
    const float* mapx;  // [H][W]
    const float* mapy;
    int B, H, W;
};
hipError_t launch_remap(const RemapArgs& a, hipStream_t s);
hipError_t launch_heat_tail(const HeatArgs& a, hipStream_t s);
hipError_t launch_heat_prob_from_raw(const HeatArgs& a, hipStream_t s);
hipError_t launch_refine(const HeatArgs& a, hipStream_t s);

// ---- line candidates: state-independent part of detectLines ----
struct LineArgs {
    const float* heat;    // refined [B][H][W]
    const float* kp_un;   // [B][max_num][2]
    const uint8_t* kp_out;
    int* counters;        // [B][8]
    int* row_cnt;         // [B][max_num+1] work: survivors per row i, then their exclusive offsets
    unsigned long long* pair_mask;  // [B][max_num][mask_words] work: probe results of row i, 64 pairs per word
    int mask_words;       // >= ceil(max_num / 256) * 4
    uint32_t* cand_ij;    // [B][cand_cap] (i << 16 | j), ascending
    float* cand_sc;       // [B][cand_cap][2] inlier rate, mean bilinear heat
    int cand_cap;
    int B, H, W, max_num;
    float heat_thresh;
    float inv_scale;
};
hipError_t launch_line_candidates(const LineArgs& a, hipStream_t s);

// ---- order-dependent tail of detectLines on the device (feature/src/PPGExtractor.cpp:315-441) ----
// Flags a frame's counters[5] may carry afterwards; any of them sends the frame to the host statement of the tail.
enum LineTailFlag { LT_GUARD = 1,   // a double-precision atan2 / sin result sat too close to a float rounding boundary
                    LT_CAP = 2 };   // more candidates, interacting pairs, edges or incident lines than the work arrays hold
struct LineTailArgs {
    const float* kp_un;       // [B][max_num][2]
    int* counters;            // [B][8]: 2 candidates, 3 key points (in); 4 edges, 5 LineTailFlag bits (out); 6 interacting pairs (work)
    const int* row_off;       // [B][max_num+1] first candidate of every row i (LineArgs::row_cnt after the scan)
    const uint32_t* cand_ij;  // [B][cand_cap]
    const float* cand_sc;     // [B][cand_cap][2]
    int cand_cap;
    float* geo;               // [B][tail_cap][3] work: length, direction seen from i, direction seen from j
    int* col_off;             // [B][max_num+1] work: candidates per second end point j, then their exclusive offsets
    int* col_cur;             // [B][max_num] work
    float* col_ent;           // [B][tail_cap][4] work: candidates grouped by j: length, direction seen from j, index (bits), pad
    uint32_t* pairs;          // [B][pair_cap][2] work: (later candidate, earlier candidate | end point << 30 | code << 31)
    int pair_cap;
    uint32_t* kill_t;         // [B][tail_cap] work
    uint32_t* pend_t;         // [B][tail_cap] work
    uint32_t* cst;            // [B][tail_cap] work: state bits per candidate
    int tail_cap;             // >= 2 * edge_cap + 2
    uint32_t* edge_idx;       // [B][edge_cap][2]
    float* edge_score;        // [B][edge_cap]
    int32_t* conn_off;        // [B][max_num+1]
    uint32_t* conn_idx;       // [B][2*edge_cap]
    int32_t* coline_off;      // [B][max_num+1]
    uint32_t* coline_pairs;   // [B][edge_cap][2]
    int edge_cap;
    int B, max_num;
    float dist_thresh, inlier_rate, heat_thresh;
};
hipError_t launch_line_tail(const LineTailArgs& a, hipStream_t s);

// ---- descriptor sampling + L2 normalisation ----
struct DescArgs {
    const float* descmap;  // [B][h][w][256]
    const int32_t* kp_xy;
    const int* counters;
    float* desc;           // [B][max_num][256]
    int B, H, W, max_num;
};
hipError_t launch_desc_sample(const DescArgs& a, hipStream_t s);

// ---- descriptor matching ----
// D[i*ldd + j] = ||A[i] - B[j]||, dim 256 (multiple of 16)
hipError_t launch_dist_matrix(const float* A, int nA, const float* B, int nB, int dim, float* D, int ldd, hipStream_t s);
// batched over blockIdx.z: A + b*sA, B + b*sB, D + b*sD, row counts nA_dev[b*sn], nB_dev[b*sn]
hipError_t launch_dist_matrix_batched(const float* A, const int* nA_dev, int maxA, size_t sA, const float* B,
                                      const int* nB_dev, int maxB, size_t sB, int dim, float* D, int ldd, size_t sD,
                                      int sn, int batch, hipStream_t s);
// distances for CSR candidate lists: Dc[cand_off[i] + k] = ||A[i] - B[cand_idx[cand_off[i]+k]]||
hipError_t launch_dist_csr(const float* A, int nA, const float* B, int dim, const int32_t* cand_off,
                           const int32_t* cand_idx, float* Dc, hipStream_t s);

struct MatchArgs {
    const float* D;            // dense: [nA][ldd]; CSR: flat
    int ldd;                   // dense row stride (0 => CSR)
    const int32_t* cand_off;   // CSR (may be nullptr when dense)
    const int32_t* cand_idx;
    const uint8_t* skipB;      // optional
    const int* nA_dev;         // optional device-resident row count (overrides nA when non-null)
    const int* nB_dev;
    int nA, nB;
    float th, ratio;
    int32_t* topk_idx;         // [nA][8] work
    float* topk_dist;          // [nA][8] work
    uint32_t* claimed;         // [(nB+31)/32] work
    int32_t* match12;          // [nA]
    float* best;               // [nA]
    float* second;             // [nA]
    int* n_matches;            // device int
    // batched launch over independent problems (dense mode only): problem b = blockIdx.y uses
    // D + b*bs_D, per-row arrays + b*bs_rows (top-k lists x8), counts nA_dev[b*bs_n], n_matches[b]
    int batch = 1;
    size_t bs_D = 0, bs_rows = 0;
    int bs_n = 0;
};
hipError_t launch_match_plain(const MatchArgs& a, hipStream_t s);

// All-pairs SearchForInitialization for `batch` independent problems of unit-norm (or all-zero) 256-float rows without
// a distance matrix: MFMA prefilter, exact chain for the listed columns, claim pass as a fixed point.  maxn <= 1024.
struct FusedMatchArgs {
    const float* A; const float* B;       // problem b: A + b*sA (rows nA_dev[b*sn]), B + b*sB (rows nB_dev[b*sn])
    size_t sA, sB;
    const int* nA_dev; const int* nB_dev; int sn;   // device-resident row counts, or both nullptr: nA, nB below (batch 1)
    int nA = 0, nB = 0;
    int maxn, batch;
    float th, ratio;
    float* norm_work;                     // [2][batch][maxn] work
    uint16_t* half_work;                  // [2][batch][maxn][256] work: the rows in fp16 (16-byte aligned)
    float* bt_work;                       // [batch][256][maxn] work: B column-major
    unsigned long long* ext_work;         // [batch][64][128] work: extended lists of the claim pass
    int* list_cnt;                        // [batch][maxn] work
    uint16_t* list_col;                   // [batch][maxn][64] work
    float* list_rad;                      // [batch][maxn] work
    int32_t* near_idx; float* near_dist;  // [batch][maxn][32] work (16-byte aligned)
    int* near_len;                        // [batch][maxn] work
    int32_t* match12;                     // [batch][maxn]
    float* best; float* second;           // [batch][maxn] or both nullptr
    int* n_matches;                       // [batch]
    int* stats = nullptr;                 // optional device ints: [0] += rounds of the claim pass, [1] += whole-row rescans,
                                          // [2..5] the slowest problem: its time and its rescan time (100 MHz ticks), rounds, rescans
    int debug_skip = 0;                   // measurement aid: 1 no claim pass, 2 no exact pass, 4 no prefilter (results are wrong)
};
hipError_t launch_match_fused(const FusedMatchArgs& a, hipStream_t s);
hipError_t launch_match_init(const MatchArgs& a, hipStream_t s);

// ---- order-dependent pass of the projection matchers ----
struct ProjMatchArgs {
    const float* D;            // distance of every listed candidate
    const int32_t* cand_off;   // [n1 + 1]
    const int32_t* cand_idx;   // slots
    int n1, n2;
    const uint8_t* blocked_in; // [n2] or nullptr
    const uint8_t* row_claims; // [n1] or nullptr
    float th, ratio;
    uint32_t* claim;           // [n2] work
    int32_t* guess;            // [n1] work
    int32_t* best_idx;         // [n1]
    float* best_dist;          // [n1]
    uint8_t* blocked_out;      // [n2] or nullptr
    int* n_matches;
};
hipError_t launch_match_projected(const ProjMatchArgs& a, hipStream_t s);

// ---- Matcher::ExtendMapMatches on the device ----
// Every array below starts on a 16-byte boundary and is readable up to the next multiple of 16 bytes (the kernel
// fetches them sixteen bytes per lane).
struct ExtendArgs {
    int m, n, n_edges, ncand;
    int n_map_edges, n_conn;       // lengths of edge_usable / edge_id and of conn_other / conn_idx
    const int32_t* cands;          // trackable map points in visiting order (sorted by edge count on the host)
    const int32_t* win_off;        // [ncand + 1] window of every candidate: key points in GetFeaturesInArea order
    const int32_t* win_idx;
    const float* win_dist;         // their descriptor distances
    const float* mdesc;            // [m][256]
    const float* fdesc;            // [n][256]
    const float* Dfull = nullptr;  // optional [m][n] distance matrix (same chain): seed weights are looked up instead of evaluated
    const uint8_t* flags;          // [m] bit 0 tracked already, bit 1 isBad(), bit 2 nObs > 0
    const int32_t* edge_off;       // [m + 1] map edges per point
    const int32_t* edge_id;
    const int32_t* edge_usable;    // per map edge: the point at its other end, -1 when the edge is bad or that point is missing
    const int32_t* conn_off;       // [n + 1] key edges per key point
    const uint32_t* conn_idx;      // their ids
    const int32_t* conn_other;     // and the key point at the other end of each
    uint8_t* tracked;              // [m] out
    int32_t* frame_mp;             // [n] in/out
    int32_t* frame_me;             // [n_edges] in/out or nullptr
    float nnratio;
    int* out;                      // [0] the reference's return value, [1] set when a list exceeded the kernel's limits
};
hipError_t launch_extend_map_matches(const ExtendArgs& a, hipStream_t s);
hipError_t launch_dist_csr_rows(const float* A, const int32_t* rows, int nrows, const float* B, const int32_t* cand_off,
                                const int32_t* cand_idx, float* Dc, hipStream_t s);

// ---- DBoW3 vocabulary tree descent ----
struct BowArgs {
    const float* desc;        // [batch][n][dim]
    size_t stride_batch;      // floats between two batches' descriptor blocks
    int n;                    // descriptors per batch entry (rows of the output)
    const int* n_dev;         // optional device-resident counts (entry b at n_dev[b*sn]); rows beyond are left untouched
    int sn;
    const float* node_desc;   // [n_nodes][dim]
    const int* child_off;     // [n_nodes + 1]
    const int* child_idx;
    const int* word_id;       // [n_nodes]
    const double* weight;     // [n_nodes]
    int dim;
    int nid_level;            // L - levelsup
    int32_t* out_word;        // [batch][n]
    double* out_weight;
    int32_t* out_node;
};
hipError_t launch_bow_transform(const BowArgs& a, int batch, hipStream_t s);

}  // namespace ppg
