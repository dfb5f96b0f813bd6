/*
 * vsf.h -- C ABI of the MI355X (gfx950) stereo feature frontend.
 *
 * This is the drop-in seam for the two private methods of slam::Frontend that call into OpenCV in
 * the reference (ut-amrl/vision_slam_frontend):
 *
 *   Frontend::ExtractFeatures  src/slam_frontend.cc:266-280  -> cv::Feature2D::detectAndCompute (ORB,
 *                              parameters :205-213) or FastFeatureDetector::detect (:271, built :191)
 *   Frontend::GetMatches       src/slam_frontend.cc:521-538  -> cv::BFMatcher(NORM_HAMMING)::knnMatch(k=2)
 *                              (:525, matcher built :247) + the ratio test (:529-536)
 *
 * Plain pointers and sizes only; every function returns a vsf_status (never aborts, never throws;
 * the reference's glog CHECK / cv::Exception / exit(1) paths become status codes).  A context owns one
 * GPU's device memory and stream and is used by one host thread at a time.  Host-pointer entry points
 * are synchronous; *_dev entry points take device pointers, are asynchronous on the context's stream (none of them waits
 * for the GPU: the one measuring call, vsf_tune_fast_resident, is explicit and says so) and are what the batched /
 * multi-GPU path uses.  A HIP failure inside an asynchronous call (a launch, an event record or wait) is returned by that
 * call as VSF_ERR_HIP -- or, if it could only be noticed later, by the next call on the context that checks (every
 * entry point that launches, and vsf_sync).
 *
 * Output record layouts are the reference's: vsf_keypoint == cv::KeyPoint (28 B), vsf_dmatch ==
 * cv::DMatch (16 B), descriptors are row-major N x 32 uint8 (cv::Mat CV_8U, 256 bit).
 */
#ifndef VSF_H_
#define VSF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSF_VERSION 2 /* 2: round 6 -- the ObserveImage queue; five options of retired kernel variants removed */
#define VSF_DESC_BYTES 32
#define VSF_MAX_LEVELS 64

typedef enum {
  VSF_OK = 0,
  VSF_ERR_INVALID_ARG = 1, /* null pointer, bad size, image geometry != context geometry */
  VSF_ERR_CAPACITY = 2,    /* an output or an internal segment overflowed; results are truncated */
  VSF_ERR_HIP = 3,         /* a HIP runtime call failed (vsf_last_hip_error gives the code) */
  VSF_ERR_UNSUPPORTED = 4, /* parameter combination outside the north-star path (e.g. WTA_K != 2) */
  VSF_ERR_NO_DEVICE = 5    /* no usable gfx950 device */
} vsf_status;

/* cv::KeyPoint */
typedef struct {
  float x, y;      /* pt, level-0 pixel coordinates */
  float size;      /* 31 * scale_l (ORB) / 7 (FAST) */
  float angle;     /* degrees, [0,360) (ORB) / -1 (FAST) */
  float response;  /* Harris response (ORB) / FAST score (FAST) */
  int32_t octave;  /* pyramid level */
  int32_t class_id;/* -1 */
} vsf_keypoint;

/* cv::DMatch */
typedef struct {
  int32_t queryIdx, trainIdx, imgIdx; /* imgIdx is always 0 */
  float distance;                     /* (float) Hamming distance */
} vsf_dmatch;

typedef struct {
  /* --- cv::ORB::create arguments (reference literals slam_frontend.cc:205-213) --- */
  int32_t nfeatures;      /* 10000 in the reference; BASELINE configs use 2000 / 8000 */
  float scale_factor;     /* 1.04f */
  int32_t nlevels;        /* 50 (<= VSF_MAX_LEVELS) */
  int32_t edge_threshold; /* 31 */
  int32_t first_level;    /* 0 (only value supported) */
  int32_t wta_k;          /* 2 (only value supported) */
  int32_t score_type;     /* 0 = cv::ORB::HARRIS_SCORE (only value supported) */
  int32_t patch_size;     /* 31 (only value supported) */
  int32_t fast_threshold; /* 20 */
  /* 1: GaussianBlur column pass rounds like OpenCV's SSE2 SymmColumnVec_32s8u (half-even) on columns
   * [0, w - w%4) and like the scalar tail (half-up) beyond; 0: half-up everywhere (non-SSE2 build). */
  int32_t blur_sse2;
  /* --- cv::FastFeatureDetector::create(10, true) (slam_frontend.cc:191), FREAK branch only --- */
  int32_t fast_detector_threshold; /* 10 */
  int32_t fast_detector_nms;       /* 1 */
  /* --- matcher: ratio test  dist1 < nn_match_ratio * dist2  with nn_match_ratio a double holding 0.6f
   * (slam_frontend.cc:523,533,555)  ==  dist1 * 2^ratio_shift < ratio_num * dist2  exactly --- */
  uint32_t ratio_num;   /* 5033165  (0.6f == 10066330 / 2^24, kept in lowest terms) */
  uint32_t ratio_shift; /* 23 */
  /* --- geometry and capacities (fixed per context) --- */
  int32_t width, height;   /* image size in pixels */
  int32_t max_images;      /* images per batch (2 per stereo frame) */
  int32_t max_keypoints;   /* per-image capacity of keypoint/descriptor outputs; 0 = nfeatures + 256 */
  /* --- RemoveAmbigStereo (slam_frontend.cc:381-383): order in which the three-term dot products of
   * left_ph.transpose() * F * right_ph are summed.  0 (default): a0*b0 + (a1*b1 + a2*b2), what Eigen 3.3's unrolled
   * reduction of a fixed-size-3 lazy product does; 1: (a0*b0 + a1*b1) + a2*b2. --- */
  int32_t residual_order;
} vsf_params;

typedef struct vsf_ctx vsf_ctx;

/* Fills *p with the reference literals for the given image size and batch capacity. */
vsf_status vsf_params_default(vsf_params* p, int width, int height, int max_images);
/* ratio as the reference stores it: a float widened to double. Sets ratio_num / ratio_shift. */
vsf_status vsf_params_set_ratio(vsf_params* p, float nn_match_ratio);

vsf_status vsf_create(const vsf_params* p, int device, vsf_ctx** out);
void vsf_destroy(vsf_ctx* ctx);
const char* vsf_status_string(vsf_status s);
/* The hipError_t of the context's last VSF_ERR_HIP -- or 10000 + the ncclResult_t when it was an RCCL call that failed
 * (vsf_comm_create, vsf_allgather_dev, vsf_gather_payload_dev).  An error belongs to the context whose call met it: one
 * noted inside an asynchronous call is returned by that call or by the same context's next call that checks, never by
 * another context driven from the same host thread. */
int vsf_last_hip_error(const vsf_ctx* ctx);
vsf_status vsf_get_params(const vsf_ctx* ctx, vsf_params* out);
/* Use an existing hipStream_t (e.g. the caller's framework stream) instead of the context's own. NULL restores it.  The
 * handle must be a live stream of the context's device (the HIP runtime does not validate stream handles). */
vsf_status vsf_set_stream(vsf_ctx* ctx, void* hip_stream);
/* Batched entry points split a batch in `lanes` halves (1 or 2, default 1) that run concurrently: one on the context's
 * stream, one on an internal stream forked from / joined back into it with events, so the caller still sees ONE
 * stream-ordered operation.  Frames are independent (slam_frontend.cc:411-416), results do not depend on it. */
vsf_status vsf_set_lanes(vsf_ctx* ctx, int lanes);
/* Cross-call pipelining for streams of batches (off by default).  With it on, the caller promises that the input
 * images of every *_batch_dev call are COMPLETE in device memory when the call is made (not merely ordered before it
 * on the stream); images produced by this library's own vsf_bayer_bg_to_gray_batch_dev on the same context are the
 * exception -- the pipelined pyramid waits for that conversion by itself.  The scale pyramid of a call -- which depends on nothing else -- is then built on internal streams
 * into the second of two pyramid buffers while the previous call's later stages are still running; all other stages
 * and all outputs stay ordered on the context's stream as before. */
vsf_status vsf_set_pipeline(vsf_ctx* ctx, int on);
/* Input readiness as an explicit event, for ANY producer (a decoder on another context or stream, the caller's own
 * kernel, a copy engine; slam_frontend_main.cc:98-132 is where frames arrive): hip_event is a hipEvent_t the caller
 * recorded behind the last operation that writes the images of the NEXT vsf_extract_batch_dev / vsf_stereo_batch_dev
 * call.  That call -- including its pipelined pyramid, which is otherwise not ordered after anything (vsf_set_pipeline)
 * -- waits for the event ON THE GPU; the host never does.  With it the promise of vsf_set_pipeline relaxes to "complete
 * when the event fires".  One-shot: the vsf_extract_batch_dev / vsf_stereo_batch_dev call that follows consumes it -- a call
 * refused with VSF_ERR_INVALID_ARG launches nothing and forgets it too (hand it over again with the corrected call) (the wait captures the event's state at that
 * moment, so the caller may record the same event again afterwards); NULL withdraws it.  The way back is ordinary
 * stream order: whatever the caller records on the context's stream after the call fires once the call has read its
 * inputs. */
vsf_status vsf_set_input_event(vsf_ctx* ctx, void* hip_event);
/* Batched entry points (>= 32 images per call) run the Gaussian blur of a call -- matrix cores and memory -- on an
 * internal stream forked behind the pyramid and joined in front of the descriptors, i.e. beside FAST and the keypoint
 * selection, which live on the vector ALU and on latency (default: on; the caller still sees ONE stream-ordered
 * operation, results do not depend on it).  With it on, the per-stage timer of the blur (vsf_profile_read) is the wall
 * span of a kernel that shares the chip, not its own duration.  0 puts the blur back in line. */
vsf_status vsf_set_blur_overlap(vsf_ctx* ctx, int on);
/* With the blur beside it, FAST can run as one RESIDENT workgroup per CU (`waves` = 2..4 waves per SIMD, fed with cells
 * through a counter) instead of a grid that fills every register of the chip for as long as cells are left, so that the
 * blur's workgroups find room beside it.  -1 (default): the form vsf_tune_fast_resident measured for this batch size, the
 * grid form for a size it has not measured; 0: always the grid form; 2..4: always resident.  Speed only: results do not
 * depend on it. */
vsf_status vsf_set_fast_resident(vsf_ctx* ctx, int waves);
/* What a batched call of the last tuned size does now: *waves = 0 (grid form) or 2..4 (resident, waves per SIMD). */
vsf_status vsf_get_fast_resident(const vsf_ctx* ctx, int* waves);
/* BLOCKING (the one *_dev-shaped call that is): runs vsf_extract_batch_dev on the given images 1 + 2 * samples times
 * (samples >= 1, 3 is a good value) -- one warm-up, then the grid form and the resident form of FAST alternately, each run
 * timed by itself between two events and waited for -- and keeps the form with the smaller MEDIAN for batches of n_images
 * images (vsf_set_fast_resident(ctx, -1) semantics).  *ms_grid / *ms_resident receive the two medians, so that the ranks of
 * a multi-GPU job can agree on one form (all-reduce the pair, then vsf_set_fast_resident with the common winner).  Batches
 * the blur does not run beside (fewer than 32 images, vsf_set_blur_overlap(ctx, 0), two lanes) are left on the grid form
 * and report 0 / 0.  No other entry point measures anything or waits for the GPU behind the caller's back. */
vsf_status vsf_tune_fast_resident(vsf_ctx* ctx, const uint8_t* d_imgs, int n_images, size_t image_stride,
                                  size_t row_stride, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts,
                                  int samples, float* ms_grid, float* ms_resident);
/* Launch choices of a context (speed / A-B measurements only: results never depend on them).  Nothing in the library
 * reads the environment.  vsf_set_option waits for the context's stream first. */
typedef enum {
  VSF_OPT_FAST_BOTH_MAX = 0, /* 16: largest batch (images) whose full and half-wave FAST cells share one launch */
  VSF_OPT_SELECT_WIDE = 1,   /* 1: a frame or two takes the 1024-thread whole-level selection class; 0: never */
  VSF_OPT_PYRAMID_FEW = 2,   /* 16: largest batch (images) whose pyramid is built by the slab kernel */
  VSF_OPT_PYRAMID_CHAIN = 3, /* 8: levels per slab launch; 0: per-level launches even for a frame or two */
  VSF_OPT_PYRAMID_ROWS = 4,  /* 6: rows of a chain's last level per slab */
  VSF_OPT_SELECT_BIG_CLASS = 5, /* 1: a batch's widest levels keep their candidate array in LDS (9 216 entries, tables in HBM) */
  VSF_OPT_PIPE_AFTER_FAST = 6, /* 1: the pipelined pyramid of a call waits for the previous call's FAST kernel; 0: it starts as soon
                                * as its inputs are ready (a step whose FAST shares the chip with a decoder) */
  VSF_OPT_PIPE_PRIORITY = 7,   /* stream priority of the pipelined pyramid chain: 0 normal, 1 lowest, -1 highest; set it before
                                * vsf_set_pipeline(ctx, 1) */
  VSF_OPT_OBSERVE_THREAD = 8,  /* 1: an ObserveImage queue of depth >= 4 gets a launcher thread -- the caller stages frames, the
                                * thread sends the batches (a host whose launches, 0.1-0.3 ms per batch, are what bounds the
                                * caller); 0 (default): the caller launches too.  Read when the queue is built */
  VSF_OPT_PYRAMID_TAIL_MIN = 9, /* smallest batch (images) whose one-band pyramid levels are one launch (a workgroup per image)
                                 * even when that fills less than three quarters of the chip; 0: never */
  VSF_OPT_OBSERVE_COPY_THREAD = 10, /* 1 (default): while frames stream into an ObserveImage queue of depth >= 4 a second host thread
                                     * takes the right image's staging copy; 0: the caller copies both.  Read when the queue is built */
  VSF_OPT_COUNT = 11
} vsf_option;
vsf_status vsf_set_option(vsf_ctx* ctx, int option, int value);
vsf_status vsf_get_option(const vsf_ctx* ctx, int option, int* value);
/* Waits for the stream and returns VSF_ERR_CAPACITY if any kernel since the last sync overflowed. */
vsf_status vsf_sync(vsf_ctx* ctx);
/* BLOCKING set-up call: sizes the scratch the batched *_dev calls keep inside the context (2-NN tables, residuals, the
 * temporal pairs' matches and sort keys, Calculate3DPoints' pair lists, the pack offsets) for batches of up to n_frames
 * stereo frames and n_pairs (past, current) pairs.  vsf_create reserves for max_images / 2 frames and as many pairs; a
 * context that serves larger batches than its own max_images suggests (the tail context of the multi-GPU composition:
 * max_images = 2, B frames, B x window pairs) calls this once.  A *_dev call beyond the reservation still works and
 * still does not wait for the GPU: it takes a new allocation (the host time of a hipMalloc) and the outgrown buffer,
 * which kernels already queued may be using, is released by the next vsf_sync / vsf_reserve / vsf_destroy. */
vsf_status vsf_reserve(vsf_ctx* ctx, int n_frames, int n_pairs);

/* Pyramid geometry the context derived (cv::ORB layer sizes / scales / per-level feature budgets). */
vsf_status vsf_level_info(const vsf_ctx* ctx, int level, int* w, int* h, float* scale, int* nfeatures);

/* ---------------- multi-GPU exchange (BASELINE configs[3]; SURVEY.md 8(b): "vsf_gather_* for multi-GPU") ----------------
 * One process (or thread) per GPU, one context and one communicator each.  What has to cross GPUs is what the reference's
 * algorithm forces: the per-frame mean residuals (the static threshold of RemoveAmbigStereo crosses frames,
 * slam_frontend.cc:353, 392-394), every rank's last frames for the temporal GetFeatureMatches (cc:424-434) and the packed
 * VisionFeature / FeatureMatch payloads one process assembles into the SLAMProblem (cc:498-503; caller
 * slam_frontend_main.cc:251, 132).  RCCL is loaded at run time by the first of these calls (VSF_ERR_UNSUPPORTED if there is
 * none); both transfers are asynchronous, ordered on the context's stream like any *_dev call. */
#define VSF_COMM_ID_BYTES 128
typedef struct vsf_comm vsf_comm;
/* ncclGetUniqueId: called by ONE rank; the caller carries the 128 bytes to the other ranks (shared memory between the
 * threads of a process, a file, MPI, the launcher's own process group ...). */
vsf_status vsf_comm_unique_id(uint8_t* id);
/* ncclCommInitRank on the context's device: collective over the `world` ranks that hold the same id. */
vsf_status vsf_comm_create(vsf_ctx* ctx, const uint8_t* id, int rank, int world, vsf_comm** out);
void vsf_comm_destroy(vsf_comm* comm);
vsf_status vsf_comm_info(const vsf_comm* comm, int* rank, int* world, int* rccl_version);
/* d_recv[r * bytes_per_rank ...] = rank r's d_send[0 .. bytes_per_rank) on every rank (ncclAllGather of bytes). */
vsf_status vsf_allgather_dev(vsf_ctx* ctx, vsf_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_rank);
/* Every rank sends `bytes` bytes of d_send to `root`; on the root rank r's bytes land at d_recv + r * recv_stride (d_recv is
 * ignored elsewhere).  One group of point-to-point transfers: over xGMI each peer -> root copy rides its own link.  `bytes`
 * is the same on all ranks (the payload sizes are exchanged first: vsf_pack_outputs_dev writes them into the header). */
vsf_status vsf_gather_payload_dev(vsf_ctx* ctx, vsf_comm* comm, const uint8_t* d_send, size_t bytes, uint8_t* d_recv,
                                  size_t recv_stride, int root);

/* ---------------- host-pointer, synchronous: one call == one reference call ---------------- */

/* detectAndCompute(image, noArray(), kps, desc)  (slam_frontend.cc:274-277).  kp_out/desc_out hold `cap`
 * records; *n_out = number of keypoints found (<= cap written). */
vsf_status vsf_extract(vsf_ctx* ctx, const uint8_t* img, int w, int h, size_t stride, vsf_keypoint* kp_out,
                       uint8_t* desc_out, int cap, int* n_out);
/* The two detectAndCompute calls of one stereo frame (slam_frontend.cc:411-412) as ONE batch of two images: one
 * upload, one set of launches, one download.  Same results as two vsf_extract calls; needs max_images >= 2. */
vsf_status vsf_extract_pair(vsf_ctx* ctx, const uint8_t* img0, const uint8_t* img1, int w, int h, size_t stride,
                            vsf_keypoint* kp0, uint8_t* desc0, int* n0, vsf_keypoint* kp1, uint8_t* desc1, int* n1,
                            int cap);
/* fast_feature_detector_->detect(image, kps)  (slam_frontend.cc:271): FAST-9/16 + NMS on the full-resolution
 * image, raster order.  threshold < 0 uses params.fast_detector_threshold. */
vsf_status vsf_fast_detect(vsf_ctx* ctx, const uint8_t* img, int w, int h, size_t stride, int threshold, int nms,
                           vsf_keypoint* kp_out, int cap, int* n_out);
/* matcher_->knnMatch(q, t, matches, 2)  (slam_frontend.cc:525): idx2/dist2 are nq x 2, ordered by
 * (distance, train index); absent neighbours are idx -1 / dist INT32_MAX. */
vsf_status vsf_knn2_hamming(vsf_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2,
                            int32_t* dist2);
/* Frontend::GetMatches(query, train, nn_match_ratio)  (slam_frontend.cc:521-538): matches in ascending
 * queryIdx.  nt < 2 yields no matches (the reference reads out of bounds there). */
vsf_status vsf_get_matches(vsf_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, vsf_dmatch* out,
                           int cap, int* n_out);

/* GetMatches of n_sets query sets against ONE train set -- the temporal loop of slam_frontend.cc:424-434, every past
 * frame against the new one -- in one upload / launch / download.  q[s] points at nq[s] rows; the matches of set s
 * are written at out + s * cap_per_set, their number to n_out[s].  Same results as n_sets vsf_get_matches calls. */
vsf_status vsf_get_matches_multi(vsf_ctx* ctx, const uint8_t* const* q, const int* nq, int n_sets, const uint8_t* t,
                                 int nt, vsf_dmatch* out, int cap_per_set, int* n_out);

/* ---------------- device-pointer, asynchronous, batched ---------------- */

/* detectAndCompute on n_images images resident in HBM.  d_imgs: image i starts at d_imgs + i*image_stride,
 * rows `row_stride` bytes apart (both multiples of 16, base 16-byte aligned).  Outputs: d_kp
 * [n_images][max_keypoints], d_desc [n_images][max_keypoints][32], d_counts [n_images]. */
vsf_status vsf_extract_batch_dev(vsf_ctx* ctx, const uint8_t* d_imgs, int n_images, size_t image_stride,
                                 size_t row_stride, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts);
/* knnMatch(k=2) + ratio test for n_pairs (query set, train set) pairs.  Set s of a descriptor array starts
 * at base + s*set_stride bytes and holds d_n*[s] rows.  pair p matches query set q_set[p] against train set
 * t_set[p] (device int32 arrays; both NULL means q_set[p] = 2p, t_set[p] = 2p+1: left vs right of stereo frame p).
 * Outputs per pair: d_idx2/d_dist2 [n_pairs][max_keypoints][2] (may be NULL), d_matches
 * [n_pairs][max_keypoints], d_nmatches [n_pairs]. */
vsf_status vsf_match_batch_dev(vsf_ctx* ctx, const uint8_t* d_desc, const int32_t* d_counts, size_t set_stride,
                               const int32_t* d_q_set, const int32_t* d_t_set, int n_pairs, int32_t* d_idx2,
                               int32_t* d_dist2, vsf_dmatch* d_matches, int32_t* d_nmatches);
/* The benchmarked hot path for a batch of stereo frames: extract(left) + extract(right) + GetMatches(left,
 * right) (slam_frontend.cc:411-416).  d_imgs is [n_frames][2][h][w]-like with the strides above (image
 * index 2f = left, 2f+1 = right). */
vsf_status vsf_stereo_batch_dev(vsf_ctx* ctx, const uint8_t* d_imgs, int n_frames, size_t image_stride,
                                size_t row_stride, vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts,
                                vsf_dmatch* d_matches, int32_t* d_nmatches);

/* ---------------- the reference's own steps between matcher and outputs, on the device ---------------- */

/* Frontend::RemoveAmbigStereo (slam_frontend.cc:353-398) for n_frames stereo frames in time order, applied to the
 * outputs of vsf_stereo_batch_dev (same layouts: image 2f = left, 2f+1 = right of frame f).  F: the fundamental
 * matrix, 9 floats row major (HOST pointer).  thr_in: the threshold in force before frame 0 (the reference's static
 * starts at 10000, cc:353).  d_thr_override: NULL, or a DEVICE array of n_frames thresholds applied as they are
 * (multi-GPU: every rank derives them from all ranks' means, vision_slam_frontend_amd/distributed.py).
 * Outputs (device): d_means [n_frames]: mean residual over ALL matches of the frame, summed in match order (NaN for
 * a frame without matches: 0/0 as in the reference, whose static is then NaN for exactly the next frame -- that frame
 * keeps nothing -- and finite again afterwards; reproduced); d_thr
 * [n_frames + 1]: threshold applied to each frame and the one in force after the batch (written without override);
 * d_kp_out / d_desc_out / d_counts_out: both frames rebuilt from the surviving pairs in match order (row i of the
 * left frame matches row i of the right frame, cc:396-397), layouts as the inputs. */
vsf_status vsf_remove_ambig_stereo_batch_dev(vsf_ctx* ctx, const vsf_keypoint* d_kp, const uint8_t* d_desc,
                                             const vsf_dmatch* d_matches, const int32_t* d_nmatches, int n_frames,
                                             const float* F, float thr_in, const float* d_thr_override,
                                             float* d_means, float* d_thr, vsf_keypoint* d_kp_out,
                                             uint8_t* d_desc_out, int32_t* d_counts_out);
/* Frontend::GetFeatureMatches (slam_frontend.cc:282-309) for n_pairs (past set, current set) pairs: GetMatches, then
 * std::sort by DMatch::operator< (distance only; the permutation is libstdc++'s), then the cut to
 * int(size * best_percent).  Set addressing as vsf_match_batch_dev.  d_pairs: [n_pairs][max_keypoints][2] uint64
 * (FeatureMatch::feature_idx_initial = index in the past set, feature_idx_current = index in the current set),
 * d_npairs [n_pairs].  Needs max_keypoints < 65536. */
vsf_status vsf_feature_matches_batch_dev(vsf_ctx* ctx, const uint8_t* d_desc, const int32_t* d_counts,
                                         size_t set_stride, const int32_t* d_q_set, const int32_t* d_t_set,
                                         int n_pairs, float best_percent, uint64_t* d_pairs, int32_t* d_npairs);

/* The three steps of vsf_remove_ambig_stereo_batch_dev as separate calls, for the multi-GPU path: a rank computes the
 * residuals and per-frame means of ITS frames, all ranks exchange the means (one float per frame), every rank derives
 * the thresholds of the whole time-ordered step and filters its own frames (slam_frontend.cc:353, 392-394).
 * vsf_stereo_residuals_batch_dev keeps the residuals inside the context for the vsf_stereo_filter_batch_dev that follows. */
vsf_status vsf_stereo_residuals_batch_dev(vsf_ctx* ctx, const vsf_keypoint* d_kp, const vsf_dmatch* d_matches,
                                          const int32_t* d_nmatches, int n_frames, const float* F, float* d_means);
/* d_means [n]: the means of n consecutive frames in time order (all ranks' frames of one step).  d_thr_state: ONE device
 * float, the threshold in force before frame 0 (initialise it to 10000, cc:353); it is advanced to the value in force
 * after frame n - 1.  d_thr [n]: thr[0] = state, thr[k] = means[k - 1] + 2. */
vsf_status vsf_stereo_thresholds_dev(vsf_ctx* ctx, const float* d_means, int n, float* d_thr_state, float* d_thr);
vsf_status vsf_stereo_filter_batch_dev(vsf_ctx* ctx, const vsf_keypoint* d_kp, const uint8_t* d_desc,
                                       const vsf_dmatch* d_matches, const int32_t* d_nmatches, int n_frames,
                                       const float* d_thr, vsf_keypoint* d_kp_out, uint8_t* d_desc_out,
                                       int32_t* d_counts_out);

/* ---------------- VisionFeature on the device ---------------- */

/* Stereo calibration as Frontend uses it (FrontendConfig, slam_frontend.cc:565-644); all row-major floats. */
typedef struct {
  float projection_left[12];   /* config_.projection_left  = K_left  * [I | 0]   (cc:595-600, 619-622) */
  float projection_right[12];  /* config_.projection_right = K_right * A_right   (cc:602-611) */
  float camera_matrix_left[9]; /* config_.camera_matrix_left (cc:575-578) */
  float distortion_left[5];    /* k1 k2 p1 p2 k3 (cc:623-628) */
  float fundamental[9];        /* config_.fundamental (cc:635-644) */
  /* Rows of the DLT system cv::triangulatePoints solves per point: 6 = OpenCV <= 3.4.1 incl. the pinned 3.2.0
   * (x*P2-P0, y*P2-P1, x*P1-y*P0 per view), 4 = OpenCV >= 3.4.2 (third row dropped).  0 means 6. */
  int32_t triangulate_rows;
} vsf_calibration;

/* slam_types::VisionFeature (slam_types.h:60-75 / VisionFeature.msg) as a 28-byte record: uint64 feature_idx (two
 * words, little endian), Vector2f pixel, Vector3f point3d. */
typedef struct {
  uint32_t feature_idx_lo, feature_idx_hi;
  float pixel[2];
  float point3d[3];
} vsf_vision_feature;
/* slam_types::FeatureMatch (slam_types.h:77-89 / FeatureMatch.msg). */
typedef struct {
  uint64_t feature_idx_initial, feature_idx_current;
} vsf_feature_match;

/* The tail of Frontend::ObserveImage (slam_frontend.cc:437-443) for n_frames frames whose left / right frames were
 * rebuilt by RemoveAmbigStereo (layouts of vsf_remove_ambig_stereo_batch_dev's outputs: set 2f = left, 2f + 1 = right):
 * Calculate3DPoints (GetFeatureMatches(right, left) with best_percent 1, cv::triangulatePoints in sorted-match order,
 * (x,y,z)/w), VisionFeature(i, keypoint i, points[i]) -- a zero point where the reference's points[i] does not exist
 * (its quirk Q5) -- and UndistortFeaturePoints (cv::undistortPoints with P = K_left).
 * d_features [n_frames][max_keypoints], d_nfeatures [n_frames]; d_npoints [n_frames] (may be NULL): triangulated points
 * per frame.  Floating point: agrees with OpenCV to rounding (1e-5 relative), not bit for bit. */
vsf_status vsf_vision_features_batch_dev(vsf_ctx* ctx, const vsf_calibration* calib, const vsf_keypoint* d_kp,
                                         const uint8_t* d_desc, const int32_t* d_counts, int n_frames,
                                         vsf_vision_feature* d_features, int32_t* d_nfeatures, int32_t* d_npoints);

/* Compact output payload of a batch (what a rank sends to rank 0): counts first, then records sized by the counts.
 *   u32 magic 'VSF1', n_frames, n_pairs, total_bytes | u32 nfeatures[n_frames] | u32 npairs[n_pairs] |
 *   vsf_vision_feature x sum(nfeatures), frame after frame | vsf_feature_match x sum(npairs), pair after pair
 * d_pairs / d_npairs: outputs of vsf_feature_matches_batch_dev ([n_pairs][max_keypoints][2] uint64); n_pairs may be 0.
 * Needs payload_cap >= vsf_packed_outputs_capacity(ctx, n_frames, n_pairs) to be safe for any counts. */
size_t vsf_packed_outputs_capacity(const vsf_ctx* ctx, int n_frames, int n_pairs);
vsf_status vsf_pack_outputs_dev(vsf_ctx* ctx, const vsf_vision_feature* d_features, const int32_t* d_nfeatures,
                                int n_frames, const uint64_t* d_pairs, const int32_t* d_npairs, int n_pairs,
                                uint8_t* d_payload, size_t payload_cap);

/* ---------------- Frontend::ObserveImage (slam_frontend.cc:400-472) as a queue of stereo frames ---------------- */

/* Everything ObserveImage computes between OdomCheck and the node / factor bookkeeping, for stereo frames given as host
 * images: upload (pinned staging), ExtractFeatures x 2 (cc:411-412), GetMatches (cc:414), RemoveAmbigStereo (cc:417, the
 * threshold lives in the context like the reference's file-static), GetFeatureMatches against the <= frame_life frames kept
 * from earlier calls (cc:424-434; their filtered descriptors stay resident in HBM), Calculate3DPoints (cc:437),
 * VisionFeature assembly + UndistortFeaturePoints (cc:438-443), one compact result per frame.  Every frame joins the window
 * and the oldest leaves once frame_life are kept (cc:467-470).
 *
 * vsf_observe_submit copies the two images into pinned staging and returns a ticket; vsf_observe_collect waits for that
 * frame and hands over its result; vsf_observe_stereo is submit + collect.  Frontend::ObserveImage returns what OdomCheck
 * decided (cc:404-409), so nothing in the reference's control flow needs a frame's result before the next frame arrives: a
 * caller may keep up to `depth` frames submitted and not collected.  Frames that wait are COALESCED: they leave for the GPU
 * as one batched extraction + one batched tail (the threshold chain and the temporal window run through the batch in frame
 * order) -- a frame's chain of ~25 launch-bound kernels costs the same whether it carries one frame or thirty.  A batch
 * leaves: when the GPU is idle and no frame has arrived for 100 us (a lone frame whose caller collects it leaves at once: the
 * synchronous call is a batch of one); when `min_batch` frames wait and fewer than `in_flight` batches are on the GPU; when
 * a full batch (max_images / 2 frames, at most `depth`) waits; or when a waiting frame is collected.  While the GPU is busy
 * or frames keep arriving, frames accumulate: the batch size follows the caller's rate.  Results are those of one frame at
 * a time, bit for bit, whatever the batches were (tests/test_gpu_observe.py).  A queue of depth >= 4 owns a host thread that
 * takes half of a streaming frame's staging copy (VSF_OPT_OBSERVE_COPY_THREAD; host memory only) and, on request, a launcher
 * thread (VSF_OPT_OBSERVE_THREAD: the caller then only stages and collects; every other entry point of the context first
 * sends what waits in the queue, so nothing ever runs beside it).
 * Tickets are collected in the order they were issued; a submit beyond `depth` uncollected frames returns
 * VSF_ERR_INVALID_ARG.  Consecutive frames with different calibrations or best_percent never share a batch; frame_life
 * changes only while the queue is empty (the window starts over).
 *
 * Result (little endian, *out_bytes bytes, at most vsf_observe_capacity()):
 *   u32 magic 'VSFO', n_pairs, nfeat, total_bytes, n_left, n_right (raw keypoints), n_stereo_matches, n_points,
 *   f32 mean residual, threshold applied, threshold in force afterwards, u32 result overflow, u32 extraction overflow
 *   (either makes the collect return VSF_ERR_CAPACITY -- for THIS frame only), 3 x u32 reserved              (64 bytes)
 *   u32 npairs[n_pairs], padded to a multiple of 4 words
 *   vsf_vision_feature x nfeat
 *   vsf_feature_match x npairs[p], p = 0 .. n_pairs-1: the temporal factors, oldest kept frame first (the order of
 *     frame_list_), and LAST the right->left matches of Calculate3DPoints in sorted order (n_pairs = kept frames + 1)
 *   vsf_keypoint x nfeat, then 32-byte descriptors x nfeat: the left frame as RemoveAmbigStereo rebuilt it (cc:396)
 * vsf_observe_reset forgets the window and puts the threshold back to 10000 (cc:353). */
size_t vsf_observe_capacity(const vsf_ctx* ctx, int frame_life);
/* depth: frames that may be submitted and not collected (0: max_images / 2; up to 1024 -- the staging and result rings are
 * pinned host memory, depth x (2 images + vsf_observe_capacity)); a batch holds min(depth, max_images / 2) frames at most.
 * min_batch (0 = a whole batch when depth >= two batches, else half the depth): while the GPU is busy, fewer waiting frames than this do not leave -- an idle GPU takes
 * whatever waits, a collect sends everything -- because a batch costs the host ~45 launches whatever it carries.  in_flight (0 = 2, at most 3): batches on the GPU at a time.  Call it before the first submit or while
 * the queue is empty; changing depth rebuilds the queue (window and threshold start over). */
vsf_status vsf_observe_configure(vsf_ctx* ctx, int depth, int min_batch, int in_flight);
vsf_status vsf_observe_stereo(vsf_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                              const vsf_calibration* calib, float best_percent, int frame_life, uint8_t* out,
                              size_t cap, size_t* out_bytes);
vsf_status vsf_observe_submit(vsf_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                              const vsf_calibration* calib, float best_percent, int frame_life, int64_t* ticket);
vsf_status vsf_observe_collect(vsf_ctx* ctx, int64_t ticket, uint8_t* out, size_t cap, size_t* out_bytes);
/* The same without the copy: *out points at the result inside the context's pinned result ring; it stays valid until `depth`
 * further frames have been submitted (the first word, the magic, reads 0 there). */
vsf_status vsf_observe_collect_view(vsf_ctx* ctx, int64_t ticket, const uint8_t** out, size_t* out_bytes);
/* Does not wait and sends nothing: *ready = 1 when the frame's result is there (its collect will not wait), 0 while it still
 * waits in staging or is on the GPU.  What a caller that books results as they come asks before each collect. */
vsf_status vsf_observe_poll(vsf_ctx* ctx, int64_t ticket, int* ready);
vsf_status vsf_observe_reset(vsf_ctx* ctx);
/* What the queue did since it was built: out[0..n) of { frames launched, batches, largest batch, batches of one frame that
 * ran on one stream, launches forced by a collect or a change of parameters, launches that had to wait for a batch slot,
 * depth, frames per batch at most, then the host's nanoseconds inside staging copies, batch launches, waits for results }. */
vsf_status vsf_observe_stats(const vsf_ctx* ctx, int64_t* out, int n);

/* SURVEY section 8(f) row f4, the decode itself: DecodeImage's cv::imdecode(msg.data, cv::IMREAD_GRAYSCALE)
 * (slam_frontend_main.cc:99-100) for n JPEG files in HOST memory (the CompressedImage payloads), all of width x height:
 * ITU-T T.81 Huffman decoding + libjpeg's ISLOW inverse DCT of the luminance component, which is what OpenCV's reader
 * computes for a gray read (JCS_GRAYSCALE; chroma is parsed and dropped).  Baseline files (SOF0 / SOF1: what a camera
 * driver writes; also when their components come in several scans) and progressive files (SOF2; any scan script that
 * brings every luminance coefficient to full precision), gray and YCbCr with sampling factors up to 4, restart intervals,
 * custom tables.  Arithmetic-coded, 12-bit and lossless files and progressive files whose scans stop short of full
 * precision (libjpeg shows an approximation of those) return VSF_ERR_UNSUPPORTED, files of another size or with malformed headers
 * VSF_ERR_INVALID_ARG (nothing is launched then).
 * The images land at d_dst + i * dst_image_stride (DEVICE memory, rows dst_row_stride apart; base and strides multiples
 * of 4) -- the input of vsf_bayer_bg_to_gray_batch_dev or of the extraction.  The files are copied before the call
 * returns; the decode is asynchronous on the context's stream (one wave per image: run it on a context / stream of its
 * own beside other work).  A call whose files do not fit the context's pinned staging and device copies (the first
 * call, or a larger batch than any before) allocates anew without waiting for the GPU; what it outgrew is released by the
 * next vsf_sync.  A stream that breaks off inside its entropy-coded data decodes as libjpeg does (zero bits) and
 * makes the next vsf_sync return VSF_ERR_INVALID_ARG.  PNG files take vsf_png_decode_gray_batch. */
vsf_status vsf_jpeg_decode_gray_batch(vsf_ctx* ctx, const uint8_t* const* jpeg, const size_t* nbytes, int n_images,
                                      int width, int height, uint8_t* d_dst, size_t dst_image_stride,
                                      size_t dst_row_stride);

/* The same for n PNG files (the other format a CompressedImage carries: image_transport's lossless setting): chunk walk and
 * CRC checks on the host (a damaged critical chunk fails the call, as it fails png_read_*; a damaged ancillary chunk is
 * skipped), RFC 1951 inflate and the PNG row filters on the device.  Every colour type, interlaced (Adam7) or not: colour type 0 at 1, 2,
 * 4, 8 and 16 bits and colour type 4 at 8 and 16 bits -- 16-bit samples keep their high byte, alpha is dropped, 1 / 2 / 4-bit
 * samples are replicated to 8 bits (grfmt_png.cpp's libpng settings for IMREAD_GRAYSCALE) --, colour types 2 and 6 at 8 and 16
 * bits and palettes of 1 to 8 bits as libpng's rgb_to_gray(1, 0.299, 0.587) makes them (the truncated / rounded integer sum,
 * or, with a gAMA outside 0.95 .. 1.05 or an sRGB chunk, the sum of the linearised samples through libpng's two tables).
 * VSF_ERR_UNSUPPORTED, never a guess: beside colour samples iCCP, more than one gAMA / sRGB, one out of range, cHRM
 * other than sRGB's primaries next to a gamma chunk, 16-bit samples with a gamma that matters.  Files of another size or with malformed
 * chunks VSF_ERR_INVALID_ARG (nothing is launched then); compressed data that breaks (what libpng answers with png_error)
 * makes the next vsf_sync return VSF_ERR_INVALID_ARG -- that includes what zlib still reads behind the image's last byte in the
 * call that delivers libpng's last row (the rest of the <= 8192-byte piece of one IDAT chunk it was fed: end-of-block code,
 * further block headers, the Adler-32) and what cv::imdecode's png_read_end makes of the rest: it drains the stream with no
 * row to fill, where zlib's errors and data behind the image are warnings, but IDAT data that runs out before the stream has
 * ended is png_error("Not enough image data") -- a file cut inside its last bytes is refused although every pixel was there.
 * Held against the real libpng driven as grfmt_png.cpp drives it (tests/png_ref.py).  Arguments and the asynchronous contract
 * as for the JPEG call. */
vsf_status vsf_png_decode_gray_batch(vsf_ctx* ctx, const uint8_t* const* png, const size_t* nbytes, int n_images,
                                     int width, int height, uint8_t* d_dst, size_t dst_image_stride,
                                     size_t dst_row_stride);

/* cv::imdecode(msg.data, cv::IMREAD_GRAYSCALE) as DecodeImage calls it (slam_frontend_main.cc:99-100), whatever the payloads
 * are: every file is told by its first bytes (JPEG: FF D8 FF, PNG: its 8-byte signature) and runs of one format go to
 * vsf_jpeg_decode_gray_batch / vsf_png_decode_gray_batch, whose rules apply; image i lands at d_dst + i * dst_image_stride.
 * A file of neither format returns VSF_ERR_UNSUPPORTED (imdecode's other decoders are not built); nothing decoded so far is
 * undone when a later run is refused. */
vsf_status vsf_imdecode_gray_batch(vsf_ctx* ctx, const uint8_t* const* files, const size_t* nbytes, int n_images,
                                   int width, int height, uint8_t* d_dst, size_t dst_image_stride,
                                   size_t dst_row_stride);

/* SURVEY section 8(f) row f4, the part behind cv::imdecode: DecodeImage's cvtColor(COLOR_BayerBG2BGR) +
 * cvtColor(COLOR_BGR2GRAY) (slam_frontend_main.cc:101-106) for n 8-bit mosaics of width x height resident in HBM, in
 * one pass.  d_src / d_dst: image i at base + i * image_stride, rows row_stride bytes apart; bases and strides multiples
 * of 4, dst_row_stride >= (width + 3) & ~3 (the padding bytes of a destination row up to that width are overwritten).
 * The result is the input of vsf_extract_batch_dev / vsf_stereo_batch_dev (whose own alignment rules apply).
 * Asynchronous on the context's stream. */
vsf_status vsf_bayer_bg_to_gray_batch_dev(vsf_ctx* ctx, const uint8_t* d_src, int n_images, int width, int height,
                                          size_t src_image_stride, size_t src_row_stride, uint8_t* d_dst,
                                          size_t dst_image_stride, size_t dst_row_stride);

/* ---------------- introspection for kernel-level parity tests and the roofline model ---------------- */

/* Test hook for the error plumbing: the next entry point that launches returns VSF_ERR_HIP (vsf_last_hip_error == code),
 * as if one of its event records / waits had failed; the call after that works again. */
vsf_status vsf_debug_inject_hip_error(vsf_ctx* ctx, int code);
/* Copies level `level` of image `image` from the last extract to host (blurred: 0 = FAST/Harris/angle input,
 * 1 = descriptor input).  out has `ostride` bytes per row. */
vsf_status vsf_debug_level_image(vsf_ctx* ctx, int image, int level, int blurred, uint8_t* out, size_t ostride);
/* FAST+NMS candidates of (image, level) in raster order after the border filter (x, y, score) -> kp_out
 * with size 7, angle -1, response = score.  */
vsf_status vsf_debug_fast_candidates(vsf_ctx* ctx, int image, int level, vsf_keypoint* kp_out, int cap,
                                     int* n_out);
/* Final per-level keypoints in level coordinates (after both retainBest cuts, with angle). */
vsf_status vsf_debug_level_keypoints(vsf_ctx* ctx, int image, int level, vsf_keypoint* kp_out, int cap,
                                     int* n_out);
/* Test hook of the order-exact selection (cv::KeyPointsFilter::retainBest as left by libstdc++'s nth_element +
 * partition): applies retainBest(n_points) on the GPU to n (key, id) pairs given as two host arrays, in place.
 * mode 0: keys are float bit patterns compared as floats; mode 1: only the top byte of the key is compared
 * (packed FAST candidates).  use_lds: run on an LDS copy when n <= 4096.  *n_out = surviving count. */
vsf_status vsf_debug_retain_best(vsf_ctx* ctx, uint32_t* key_bits, uint32_t* ids, int n, int n_points, int use_lds,
                                 int mode, int* n_out);
/* Test hook of the order-exact sort of Frontend::GetFeatureMatches (slam_frontend.cc:289-291): n_lists lists of n host
 * matches each (list i at matches + i * n; the context's max_keypoints bounds n) go through the device's std::sort
 * restatement + the cut to int(n * best_percent); pairs_out[i * n * 2 ...] receives (queryIdx, trainIdx) of the survivors
 * in sorted order and counts_out[i] their number.  serial != 0 forces the one-lane kernel. */
vsf_status vsf_debug_sort_trim(vsf_ctx* ctx, const vsf_dmatch* matches, int n_lists, int n, float best_percent,
                               int serial, uint64_t* pairs_out, int32_t* counts_out);

/* Test hook of the JPEG ingest: on != 0 sends every file through the one-wave-per-image decoder (which otherwise takes the
 * files with restart intervals and, a second time, damaged progressive files), so that the parallel forms can be held
 * against it file by file.  Waits for the context's stream. */
vsf_status vsf_debug_jpeg_serial(vsf_ctx* ctx, int on);

/* Per-stage device timing (hipEvents recorded on the context's stream around every stage of the batched entry
 * points).  vsf_profile_read synchronises the stream, adds the elapsed milliseconds and launch counts of every
 * stage executed since the last reset into ms_total[] / launches[] (VSF_STAGE_COUNT entries each). */
#define VSF_STAGE_COUNT 8
enum { VSF_STAGE_PYRAMID = 0, VSF_STAGE_FAST, VSF_STAGE_SELECT, VSF_STAGE_BLUR, VSF_STAGE_DESCRIBE, VSF_STAGE_KNN2,
       VSF_STAGE_RATIO, VSF_STAGE_TAIL /* residuals, thresholds, filter, sort + trim, 3-D points, pack */ };
vsf_status vsf_profile_enable(vsf_ctx* ctx, int on);
vsf_status vsf_profile_read(vsf_ctx* ctx, double* ms_total, int64_t* launches, int reset);
const char* vsf_stage_name(int stage);

/* Algorithmic HBM bytes of one extract of one image: SURVEY.md section 8(d) B_img for this geometry. */
uint64_t vsf_algorithmic_bytes_per_image(const vsf_ctx* ctx);
/* Total pyramid pixels P = sum_l w_l*h_l. */
uint64_t vsf_pyramid_pixels(const vsf_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* VSF_H_ */
