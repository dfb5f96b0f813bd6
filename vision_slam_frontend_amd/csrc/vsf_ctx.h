// vsf_ctx.h -- the context behind the C ABI (include/vsf.h) and the helpers its entry points share.  Private to the
// library: vsf_geometry.hip builds the per-size tables, vsf_api.hip owns creation / options / scratch and the two
// composed stages (extract_on, match_on), vsf_batch.hip the batched *_dev entry points, vsf_observe.hip the ObserveImage
// queue, vsf_host.hip the host-pointer calls, vsf_ingest.hip the decoders' entry points, vsf_debug.hip the test hooks.
#ifndef VSF_CTX_H_
#define VSF_CTX_H_

#include <cstddef>
#include <cstdint>
#include <vector>

#include "vsf_internal.h"

namespace vsfi {

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

struct Geometry {
  VsfGeom g{};
  std::vector<VsfLevel> levels;
  std::vector<uint32_t> units;
  std::vector<VsfTap> xt, yt;
  // matrix-core blur (k_blur.hip blur_mma_kernel): work units and constant MFMA operands
  std::vector<uint32_t> blur_mma_units, blur_mma_units_small;  // long strips (batches) / short strips (a frame or two)
  std::vector<uint4> blur_tcol, blur_tv;
  int blur_bias = 0;
};

struct DevSet {  // device copies of one Geometry + its work buffers
  VsfDev d{};
  VsfLevel* levels = nullptr;
  uint32_t* units = nullptr;
  uint32_t* blur_mma_units = nullptr;
  uint32_t* blur_mma_units_small = nullptr;
  uint4* blur_tcol = nullptr;
  uint4* blur_tv = nullptr;
  uint2* ic_table = nullptr;
  bool ready = false;
};

// vsf_geometry.hip
bool build_geometry(const vsf_params& p, bool orb, bool nms, Geometry* out);
void gaussian_taps(int k[4]);
std::vector<uint2> build_ic_table();
std::vector<int> orb_umax(int patch_size);

}  // namespace vsfi

struct vsf_ctx {
  vsf_params p{};
  int device = 0;
  int n_cus = 256;
  hipStream_t own_stream = nullptr, stream = nullptr;
  // Second lane of the batched entry points: half of a batch runs on `stream`, the other half on `aux_stream`
  // (frames are independent), so latency-bound stages of one half overlap VALU-bound stages of the other.
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // the blur (matrix cores + memory) beside FAST (vector ALU) in batched calls: its own stream, forked after the pyramid
  hipStream_t blur_stream = nullptr;
  hipEvent_t ev_blur_fork = nullptr, ev_blur_done = nullptr;
  VsfSideStream side{};  // aux_stream, for the pyramid's second launch chain
  // Cross-call pipelining (vsf_set_pipeline): the pyramid of call k + 1 is built on side streams, into the other of
  // two pyramid buffers, while call k's later stages still run.
  bool pipeline = false;
  hipStream_t pipe_stream = nullptr;  // the pipelined chain's own stream when VSF_OPT_PIPE_PRIORITY asks for a priority
  int pipe_stream_priority = 0;
  uint8_t* pyr_alt = nullptr;
  int pyr_flip = 0;
  hipEvent_t ev_pyr_done = nullptr, ev_pyr_free[2] = {nullptr, nullptr}, ev_fast_done = nullptr;
  bool pyr_free_valid[2] = {false, false}, fast_done_valid = false;
  // A producer the library owns (the Bayer ingest) records this on the context's stream; a pipelined pyramid, which is
  // NOT ordered after that stream's earlier work, waits for it.
  hipEvent_t ev_ingest_done = nullptr;
  bool ingest_done_valid = false;
  // ... and ANY other producer hands over an event of its own (vsf_set_input_event): the next batched call -- its
  // pipelined pyramid included -- waits for it; one-shot.
  hipEvent_t input_event = nullptr;
  const uint8_t* last_pyr = nullptr;
  int lanes = 1;  // 1 = everything on `stream` (default), 2 = two concurrent half batches (vsf_set_lanes)
  int blur_overlap = 1;  // the blur on blur_stream beside FAST / selection (vsf_set_blur_overlap)
  int fast_resident = -1;  // vsf_set_fast_resident
  int fast_force = -1;     // vsf_tune_fast_resident only: the form of the run it is timing
  VsfTuning tuning;        // vsf_set_option
  int last_hip = 0;
  int pending_hip = 0;  // an error noted during one of THIS context's calls that returned before checking (VsfErrorScope)
  vsfi::Geometry orb, fast;
  vsfi::DevSet dorb, dfast;
  // Status words (bit 0: capacity overflow, bit 1: a JPEG stream broke off): word 0 belongs to the context's own stream
  // (batched and host-pointer calls, vsf_sync), words 1..6 to the frames that may be in flight (vsf_observe_submit) --
  // a frame's kernels run on its slot's stream beside another frame's, so each frame sets, copies and clears its own word.
  int32_t* d_status = nullptr;     // [1 + VSF_OBSERVE_MAX_SLOTS]
  uint32_t* fast_cells = nullptr;  // [2] cell counters of the resident FAST kernels (k_fast.hip)
  struct FastTune {  // resident FAST or one workgroup per four cells: what vsf_tune_fast_resident measured, per batch size
    int n = 0, choice = -1;
    hipEvent_t ev[2] = {nullptr, nullptr};
  } fast_tune;
  int32_t* h_status = nullptr;  // pinned
  // staging for the host-pointer entry points
  uint8_t* st_img = nullptr;
  size_t st_img_pitch = 0, st_img_stride = 0;
  vsf_keypoint* st_kp = nullptr;
  uint8_t* st_desc = nullptr;
  int32_t* st_counts = nullptr;
  // matcher work buffers
  int32_t* m_idx2 = nullptr;
  int32_t* m_dist2 = nullptr;
  int m_pairs = 0, m_rows = 0;
  // f1 work buffers: residuals [frames][rows], F (9 floats), matches / counts / sort keys of the temporal pairs
  float* f_residual = nullptr;
  int f_frames = 0;
  vsf_dmatch* t_matches = nullptr;
  int32_t* t_nmatches = nullptr;
  void* t_sortkeys = nullptr;
  int t_pairs = 0;
  // f2 work buffers: right->left pairs of every frame, their set indices, the pack kernel's offsets
  uint64_t* v_pairs = nullptr;
  int32_t* v_npairs = nullptr;
  int32_t* v_sets = nullptr;   // [2][v_frames]: q_set = 2f + 1, t_set = 2f
  int v_frames = 0;
  uint32_t* pk_offsets = nullptr;
  int pk_entries = 0;
  // Scratch a *_dev call has outgrown.  Such a call takes a NEW allocation (hipMalloc does not wait for the GPU) and
  // parks the old one here, because hipFree would wait for the whole device behind the caller's back; released by
  // vsf_sync / vsf_reserve / vsf_destroy, when every stream of the context is known to be idle.
  std::vector<void*> retired;
  std::vector<void*> retired_host;  // ... and pinned host buffers (the ingest's staging)
  // The ObserveImage queue (vsf_observe.hip): frames wait in pinned staging and leave for the GPU in batches.
  static constexpr int kObserveBatchSlots = 4;
  struct ObserveLauncher;    // the queue's lock and its launcher thread
  struct ObserveCopyHelper;  // a host thread that takes half of a frame's staging copy while frames stream in
  struct ObserveBatchMeta;  // pinned, device-visible: read by the kernels over PCIe (no copy command)
  struct ObserveBatch {     // what one batch's extraction writes and its tail reads
    uint8_t* d_img = nullptr;        // [2 bmax] images at the staging pitch
    vsf_keypoint* kp_raw = nullptr;  // [2 bmax][K]
    uint8_t* desc_raw = nullptr;     // [2 bmax][K][32]
    int32_t* counts_raw = nullptr;   // [2 bmax]
    vsf_dmatch* matches = nullptr;   // [bmax][K] raw stereo matches
    int32_t* nmatches = nullptr;     // [bmax]
    int32_t* status = nullptr;       // [2 bmax] a status word per image
    ObserveBatchMeta* h_meta = nullptr;
    hipEvent_t ev_uploaded = nullptr, ev_extracted = nullptr, ev_done = nullptr;
    bool used = false;               // ev_done has been recorded at least once
    hipStream_t done_stream = nullptr;  // the stream its tail ran on
  };
  struct ObserveFrame {  // per frame slot (ticket % depth), host side
    vsf_calibration calib;
    float best_percent = 0.f;
    int batch = -1;  // batch slot it was launched in, -1 while it waits
  };
  struct Observe {
    bool ready = false;
    int frame_life = 0;
    int depth = 0;      // frames that may be submitted and not collected
    int bmax = 0;       // frames per batch at most
    int ring = 0;       // descriptor sets [0, ring): the kept left frames (frame g in set g % ring); [ring, ring + bmax): the
                        // right frames of the batch in the tail
    int max_pairs = 0;  // bmax * (frame_life + 1)
    uint8_t* sets = nullptr;        // [ring + bmax][K][32]
    int32_t* set_counts = nullptr;  // [ring + bmax]
    // the tail's scratch exists once: tails run one after the other (they carry the threshold and the window)
    float* residual = nullptr;      // [bmax][K]
    float* floats = nullptr;        // means [bmax] | thr [bmax + 1] | thr_state
    vsf_keypoint* kpf = nullptr;    // [2 bmax][K]
    int32_t* ints = nullptr;        // counts_f [2 bmax] | nfeat [bmax] | npoints [bmax]
    int32_t *ex_idx2 = nullptr, *ex_dist2 = nullptr;  // [bmax][K][2] the extraction side's matcher scratch
    int32_t *t_idx2 = nullptr, *t_dist2 = nullptr;    // [max_pairs][K][2] the tail's
    vsf_dmatch* t_matches = nullptr;                  // [max_pairs][K]
    int32_t* t_nmatches = nullptr;
    void* t_sortkeys = nullptr;
    uint64_t* pairs = nullptr;      // [max_pairs][K][2]
    int32_t* npairs = nullptr;
    vsf_vision_feature* features = nullptr;  // [bmax][K]
    uint8_t* h_img = nullptr;       // pinned [depth][2] images at the staging pitch
    uint8_t* h_out = nullptr;       // pinned [depth][out_stride], written by observe_pack_kernel
    size_t out_cap = 0, out_stride = 0;
    ObserveBatch batch[kObserveBatchSlots];
    std::vector<ObserveFrame> frames;  // [depth]
    hipStream_t copy_stream = nullptr, tail_stream = nullptr;
    ObserveLauncher* launcher = nullptr;
    ObserveCopyHelper* copy_helper = nullptr;
    int64_t next_ticket = 0;   // tickets issued
    int64_t next_launch = 0;   // first frame still waiting in staging
    int64_t next_collect = 0;  // oldest frame not collected
    int64_t batches = 0;       // batches launched
    int64_t last_submit_ns = 0;  // when the last frame arrived
    int rows_hint = 0;           // expected rows of a filtered frame (from the collected results; 0: unknown)
    int last_batch = -1;       // slot of the batch launched last
    int64_t stat_frames = 0, stat_max_batch = 0, stat_solo = 0, stat_forced = 0, stat_slot_waits = 0;  // vsf_observe_stats
    int64_t stat_copy_ns = 0, stat_launch_ns = 0, stat_wait_ns = 0;  // host time in staging copies, launches, waits
  } ob;
  // vsf_observe_configure (before the queue is built by the first submit; 0 = defaults)
  int ob_depth = 0, ob_min_batch = 0, ob_in_flight = 2;
  // vsf_jpeg_decode_gray_batch: pinned staging + device copy of the packed headers / tables / entropy-coded segments
  // (two sets, used alternately: the host fills one while the previous call's upload / decode still use the other)
  int32_t* jp_flags = nullptr;  // [jp_flags_cap] per progressive file of a call: damaged, decode again scan after scan
  int jp_flags_cap = 0;
  uint8_t* jp_host[2] = {nullptr, nullptr};
  uint8_t* jp_dev[2] = {nullptr, nullptr};
  size_t jp_cap[2] = {0, 0};
  hipEvent_t jp_copied[2] = {nullptr, nullptr};  // the last upload out of jp_host[i] has finished
  int jp_flip = 0;
  uint8_t* png_filtered = nullptr;  // PNG: the inflated scanlines of a batch
  size_t png_filtered_cap = 0;
  int32_t* png_file_status = nullptr;
  int png_file_status_cap = 0;
  uint8_t* jp_clean = nullptr;   // parallel decode: the de-stuffed streams (layout of the upload's stream part)
  size_t jp_clean_cap = 0;
  int16_t* jp_coef = nullptr;    // ... and the luminance coefficients of the batch
  size_t jp_coef_cap = 0;
  uint8_t* mh_desc = nullptr;  // host-API descriptor staging: 2 sets
  int32_t* mh_counts = nullptr;
  vsf_dmatch* mh_matches = nullptr;
  int32_t* mh_nmatches = nullptr;
  int mh_rows = 0;
  // vsf_get_matches_multi staging: sets x rows descriptors, per-set counts / set indices / matches
  uint8_t* mm_desc = nullptr;
  int32_t* mm_counts = nullptr;  // [sets + 1] counts, then [sets] q_set, [sets] t_set
  vsf_dmatch* mm_matches = nullptr;
  int32_t* mm_nmatches = nullptr;
  int mm_sets = 0, mm_rows = 0;
  VsfImages last_images{};
  bool last_valid = false;
  bool fast_nms = true;  // NMS mode the standalone-FAST geometry was built for
  // per-stage hipEvent profiling
  bool prof_on = false;
  std::vector<hipEvent_t> ev_pool;  // pairs
  std::vector<int> ev_stage;        // stage of pair i
  std::vector<int> ev_launches;
  size_t ev_used = 0;               // pairs in flight
  double prof_ms[VSF_STAGE_COUNT] = {0};
  int64_t prof_launches[VSF_STAGE_COUNT] = {0};
};

#define VSF_HIP(call)                     \
  do {                                    \
    hipError_t e_ = (call);               \
    if (e_ != hipSuccess) {               \
      ctx->last_hip = (int)e_;            \
      return VSF_ERR_HIP;                 \
    }                                     \
  } while (0)
// End of an entry point that launched: a failed launch (hipGetLastError) or anything a launcher / stream helper noted
// (vsf_note: event records and waits, memsets) becomes this call's VSF_ERR_HIP.
#define VSF_STICKY()                                               \
  do {                                                             \
    hipError_t e_ = hipGetLastError();                             \
    if (e_ == hipSuccess) e_ = (hipError_t)vsf_tls_hip_error;      \
    if (e_ == hipSuccess) e_ = (hipError_t)ctx->pending_hip;       \
    vsf_tls_hip_error = 0;                                         \
    ctx->pending_hip = 0;                                          \
    if (e_ != hipSuccess) {                                        \
      ctx->last_hip = (int)e_;                                     \
      return VSF_ERR_HIP;                                          \
    }                                                              \
  } while (0)

namespace vsfi {

template <class T>
hipError_t upload(T** dst, const std::vector<T>& v) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), v.size() * sizeof(T));
  if (e != hipSuccess) return e;
  return hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

vsf_status alloc_devset(vsf_ctx* ctx, const Geometry& G, DevSet* ds, bool orb, int n_images);
void free_devset(DevSet* ds);

// ---- scratch that follows the batch size of the *_dev calls ----
// Sized at vsf_create for max_images / 2 frames and as many pairs, or by vsf_reserve.  A call that needs more never waits
// for the GPU: grow_scratch() allocates anew and retires the old buffer (kernels already queued keep using it).
template <class T>
vsf_status grow_scratch(vsf_ctx* ctx, T*& ptr, size_t bytes) {
  void* fresh = nullptr;
  VSF_HIP(hipMalloc(&fresh, std::max<size_t>(bytes, 16)));
  if (ptr) ctx->retired.push_back(static_cast<void*>(ptr));
  ptr = static_cast<T*>(fresh);
  return VSF_OK;
}
void free_retired(vsf_ctx* ctx);  // (callers have waited for every stream of the context)
vsf_status ensure_match_buffers(vsf_ctx* ctx, int pairs, int rows);
vsf_status ensure_match_host_staging(vsf_ctx* ctx, int rows);  // (host-pointer, synchronous entry points only)
vsf_status ensure_residual_buffers(vsf_ctx* ctx, int n_frames);
vsf_status ensure_temporal_buffers(vsf_ctx* ctx, int n_pairs);
vsf_status ensure_vision_buffers(vsf_ctx* ctx, int n_frames);
vsf_status ensure_pack_buffers(vsf_ctx* ctx, int n);
vsf_status reserve_scratch(vsf_ctx* ctx, int n_frames, int n_pairs);
vsf_status ensure_pipeline_buffers(vsf_ctx* ctx);
void free_observe(vsf_ctx* ctx);          // vsf_observe.hip
void stop_observe_threads(vsf_ctx* ctx);  // ... before anything waits for the context's streams to drain

vsf_status check_status_word(vsf_ctx* ctx);
vsf_status validate_images(const vsf_ctx* ctx, const uint8_t* d_imgs, int n, size_t image_stride, size_t row_stride);
void prof_fold(vsf_ctx* ctx);         // stream must be idle
void sync_all_streams(vsf_ctx* ctx);  // every stream the context launches on

struct StageTimer {  // records an event pair around one stage when profiling is on
  vsf_ctx* ctx;
  size_t slot = 0;
  bool on;
  hipStream_t st;
  StageTimer(vsf_ctx* c, hipStream_t stream, int stage, int launches) : ctx(c), on(c->prof_on), st(stream) {
    if (!on) return;
    if (ctx->ev_used >= 2048) {
      sync_all_streams(ctx);
      prof_fold(ctx);
    }
    slot = ctx->ev_used++;
    while (ctx->ev_pool.size() < 2 * (slot + 1)) {
      hipEvent_t e = nullptr;
      vsf_note(hipEventCreate(&e));
      ctx->ev_pool.push_back(e);
    }
    if (ctx->ev_stage.size() <= slot) {
      ctx->ev_stage.resize(slot + 1);
      ctx->ev_launches.resize(slot + 1);
    }
    ctx->ev_stage[slot] = stage;
    ctx->ev_launches[slot] = launches;
    vsf_note(hipEventRecord(ctx->ev_pool[2 * slot], st));
  }
  ~StageTimer() {
    if (on) vsf_note(hipEventRecord(ctx->ev_pool[2 * slot + 1], st));
  }
};

// The per-image work buffers of images [i0, i0 + n) seen as a batch of their own.
VsfDev shifted(const VsfDev& d, const VsfGeom& g, int i0);

// vsf_set_input_event: the batched call that follows waits for the caller's event on the context's stream (level 0 of the
// pyramid IS the input: FAST, Harris and the orientation read it there) -- the pipelined pyramid chain waits for it by
// itself in extract_on -- and the event is forgotten when the call returns (one-shot).
struct InputEventScope {
  vsf_ctx* ctx;
  // Built FIRST in the entry point (right behind the null check), so that every way out -- a refusal included -- forgets the
  // event: a handle left pending would be waited for by some later call, when the caller may long have destroyed or
  // re-recorded it.  wait() is issued once the arguments have been validated.
  explicit InputEventScope(vsf_ctx* c) : ctx(c) {}
  void wait() {
    if (ctx->input_event) vsf_note(hipStreamWaitEvent(ctx->stream, ctx->input_event, 0));
  }
  ~InputEventScope() { ctx->input_event = nullptr; }
};

// detectAndCompute for images [i0, i0 + n) of `im` on stream `st`.  `status`: the status word the kernels report capacity
// overflows into (the context's, or the word of the frame in flight that owns this extraction).
void extract_on(vsf_ctx* ctx, hipStream_t st, const VsfImages& im_all, int i0, int n, vsf_keypoint* d_kp, uint8_t* d_desc,
                int32_t* d_counts, bool inputs_complete = false, const VsfSideStream* own_side = nullptr,
                int32_t* status = nullptr, int status_stride = 0);
// knnMatch(k = 2) + ratio test for pairs [p0, p0 + n) on stream `st`.
void match_on(vsf_ctx* ctx, hipStream_t st, const uint8_t* d_desc, const int32_t* d_counts, size_t set_stride,
              const int32_t* d_q_set, const int32_t* d_t_set, int p0, int n, int32_t* d_idx2, int32_t* d_dist2,
              vsf_dmatch* d_matches, int32_t* d_nmatches, int32_t* status = nullptr);
vsf_status fork_lane(vsf_ctx* ctx);
vsf_status join_lane(vsf_ctx* ctx);

// Runs body(stream, first, count) over `units` work items (images or stereo frames): all on the context's stream, or
// (vsf_set_lanes(ctx, 2)) as two halves on the two lanes.  Measured on MI355X: the stages are either VALU-bound
// (FAST, blur) or latency-bound, and a VALU-bound kernel at full occupancy leaves no registers for a second kernel's
// waves, so the second lane only fills launch gaps and tails (+5 % frames/s) while every kernel's own duration
// roughly doubles; one lane stays the default.
template <class Body>
vsf_status run_chunked(vsf_ctx* ctx, int units, Body body) {
  if (ctx->lanes < 2 || units < 2) {
    body(ctx->stream, 0, units);
    return VSF_OK;
  }
  vsf_status st = fork_lane(ctx);
  if (st != VSF_OK) return st;
  const int n0 = (units + 1) / 2;
  body(ctx->stream, 0, n0);
  body(ctx->aux_stream, n0, units - n0);
  return join_lane(ctx);
}

vsf_status extract_async(vsf_ctx* ctx, const VsfImages& im, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts,
                         bool inputs_complete = false);

}  // namespace vsfi

#endif  // VSF_CTX_H_
