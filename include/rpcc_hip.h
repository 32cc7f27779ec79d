/*
 * rpcc_hip.h -- C ABI of librpcc_hip.so: the MI355X (gfx950) implementation of the R-PCC per-frame
 * compression hot path.  Plain pointers and sizes only; every pointer marked "dev" is a device
 * (HBM) pointer, every kernel is enqueued on the caller's hipStream_t (passed as void*) and the call
 * returns without synchronising unless stated.  The library allocates nothing: work buffers come
 * from the caller (rpcc_workspace_bytes).  Return value: 0 = OK, negative = error
 * (rpcc_last_error() gives the text); nothing throws or exits across this boundary.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository).  A "frame" is one LiDAR sweep; a batch holds B frames of one lidar geometry (H x W
 * range image, P = H*W pixels); M = cluster_num (cfgs/compressor.yaml:22); labels are
 * 0 = ground, 1 = empty pixel, 2..M+1 = FPS cluster k-2 (utils/segment_utils.py:168-169), K = M+2.
 */
#ifndef RPCC_HIP_H
#define RPCC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPCC_OK 0
#define RPCC_ERR_ARG (-1)
#define RPCC_ERR_HIP (-2)
#define RPCC_MAX_CLUSTERS 254 /* labels are stored as uint8 (more clusters: the rpcc_*_wide entries below) */
#define RPCC_MAX_BATCH 65535   /* frames per call: the frame index is a grid dimension */
#define RPCC_INFO_INTS 8       /* int32 per frame in the `info` arrays below */
#define RPCC_FPS_BRUTEFORCE 1  /* flag: farthest point sampling by the one-pass-per-centre kernel (test reference) */
/* The two things of the reference's CUDA binary that its source leaves to the compiler / the reduction tree (no golden vector
 * exists for either; measured effect on the selected centres: profiles/r03_fps_mode_sensitivity.md -- none in 305 frames).
 * Default (no flag): un-fused distance, lowest index among exactly equal values.  A mode flag runs the reference's own
 * algorithm (one pass per centre), about 16x the time of the pruned kernel. */
#define RPCC_FPS_FMA1 2        /* distance of sampling_gpu.cu:64 contracted as nvcc --fmad=true may: fma(dz,dz,fma(dx,dx,dy*dy)) */
#define RPCC_FPS_FMA2 4        /* ... or fma(dz,dz,fma(dy,dy,dx*dx)) */
#define RPCC_FPS_TIE_CUDA 8    /* winner among exactly equal values = the survivor of the kernel's shared-memory tree
                                  (sampling_gpu.cu:16-21,55-69,74-134): smallest bit-reversed (k mod block), then smallest k */

/* Re-entrancy: the library keeps no per-call state of its own (a cache of kernel attributes already set, behind a
 * mutex, is all it holds); every call works on the caller's buffers and stream, so host threads may call concurrently
 * on different streams with different buffers (the reference's ThreadPoolExecutor front-end,
 * tools/compress_datalist.py:202-206).  The one developer hook that is process-wide is rpcc_debug_stamps. */

/* Interface version: changes whenever the layout of a struct below or the meaning of an argument changes (the structs carry no size
 * field).  A binding compares rpcc_version() with the RPCC_ABI_VERSION of the header it was built against before it calls anything else
 * (r-pcc_amd/_lib.py does).  100: round 3.  101: rpcc_batch_io.point_stride_bytes.  102: the uint16-label entries (rpcc_*_wide), rpcc_compress_batch_stages.
 * 103: rpcc_project_ordered, the RPCC_PROJECT_* bits of rpcc_batch_io.flags, the uint16 stage entries (rpcc_assign_wide ...). */
#define RPCC_ABI_VERSION 103
int rpcc_version(void);
const char *rpcc_last_error(void);

/* Lidar geometry: the scalars PCTransformer.__init__ derives from a lidar YAML
 * (dataset/transformer.py:26-37); the three angles are the python doubles narrowed to C float
 * exactly as pybind11 does at cpp_modules.cpp:427-428. */
typedef struct rpcc_geom {
    int32_t H, W;
    float horizontal_fov, vertical_max, vertical_min; /* radians */
} rpcc_geom;

/* ---- a2: spherical projection ------------------------------------------------------------- *
 * replaces dataset_utils_cpp.point_cloud_to_range_image_even (cpp_modules.cpp:427-467), batched.
 *   xyz      dev f32 [total,3]   points of all frames back to back (AoS, as np.fromfile gives)
 *   offsets  dev i64 [B+1]       frame b owns points offsets[b]..offsets[b+1]
 *   total    host                offsets[B]
 *   ri       dev f32 [B,P]  out  min positive depth per pixel, 0 where empty
 *   scratch  dev, scratch_bytes  work buffer (contents undefined on return).  With
 *            rpcc_project_scratch_bytes(total,B,P) bytes the LDS-band path runs (about 40 bytes of address
 *            space per point for images of up to four 32768-pixel bands -- the records are binned by
 *            chunk of 2048 points and band in fixed shares, 6 bytes per point are touched); with at least
 *            B*(P+8)*4 bytes, or an image of more than eight bands, the device-atomic path runs (same result).
 * Exact reference semantics incl. a depth-0 point resetting its pixel in input order.  Points
 * whose depth is not finite are skipped (reference: undefined behaviour).
 * The pixel of a point is first computed by a screened fast path and recomputed with the exact
 * operation sequence whenever it is not provably the same (DESIGN.md "Projection").
 * rpcc_project_fastpath_check (test hook): counts dev u64[5] = {points the fast path is certain
 * about, of those the ones whose pixel differs from the exact sequence (must be 0), points sent
 * to the exact sequence, largest |fast - exact| of the pre-rounding column and row coordinate in
 * units of 1e-9 over the points of ordinary magnitude}. */
size_t rpcc_project_scratch_bytes(int64_t total, int B, int P);
int rpcc_project_fastpath_check(const float *xyz, int64_t total, rpcc_geom g, uint64_t *counts, void *stream);
int rpcc_project(const float *xyz, const int64_t *offsets, int64_t total, int B, rpcc_geom g, float *ri,
                 void *scratch, size_t scratch_bytes, void *stream);
/* The same on the points AS STORED: the reference reads a sweep as np.fromfile(path, float32).reshape(-1, 4) -- rows
 * (x, y, z, intensity) -- and slices [:, :3] on the host (dataset/dataset.py:48-50,62).  With point_stride_bytes = 16 the
 * rows go to the device as they are (file -> pinned buffer -> DMA, no host pass over the points) and the kernel reads one
 * 16-byte row per point (points 16-byte aligned); 12 (or 0) = packed xyz = rpcc_project.  offsets / total count POINTS. */
int rpcc_project_strided(const float *points, int point_stride_bytes, const int64_t *offsets, int64_t total, int B,
                         rpcc_geom g, float *ri, void *scratch, size_t scratch_bytes, void *stream);
/* Sweeps in SCANNER ORDER (opt-in).  The reference's inputs are .bin files as the scanner wrote them (dataset/dataset.py:48-50): ring after ring, so
 * consecutive points fall into neighbouring rows of the range image.  With RPCC_PROJECT_ORDER_PROBE one workgroup per frame first PROBES the order
 * of the frame's points (16 runs of 64 points); a frame whose points move through the image ring by ring is projected by one kernel that keeps a
 * window of image rows in LDS and never writes a per-point record (csrc/project_ordered.h), any other frame -- shuffled points, as the synthetic
 * benchmark sweeps -- by the two record kernels.  Both give the reference's image bit for bit for ANY order (the window kernel re-opens rows it has
 * already written when a late point asks for one); the probe only chooses.  Off by default: measured on MI355X the window kernel -- one
 * 1024-thread workgroup per frame, 4 wavefronts per SIMD -- takes 290 us per 256 stored sweeps against 236 us of the two record kernels, and the
 * probe launch costs the shuffled benchmark 1.3 % (profiles/HISTORY.md, round 6).  order_flags / rpcc_batch_io.flags:
 *   RPCC_PROJECT_ORDER_PROBE      probe every frame's point order, window kernel for the frames that pass
 *   RPCC_PROJECT_FORCE_ORDERED    test hook: every frame with a point through the window kernel, whatever its order
 * rpcc_project_ordered = rpcc_project_strided with those flags and, in `accepted` (dev i32 [B], may be NULL), which frames the window kernel took.
 * Images the window kernel does not take (more than 128 rows, a width that is no multiple of four or above 16384) are never probed. */
#define RPCC_PROJECT_ORDER_PROBE 16
#define RPCC_PROJECT_FORCE_ORDERED 32
int rpcc_project_ordered(const float *points, int point_stride_bytes, const int64_t *offsets, int64_t total, int B,
                         rpcc_geom g, float *ri, void *scratch, size_t scratch_bytes, int order_flags, int32_t *accepted, void *stream);

/* ---- a4: ground plane ----------------------------------------------------------------------- *
 * replaces the ground branch of PointCloudSegment.segment: candidate selection + RANSAC
 * (utils/segment_utils.py:74-82,101-108).  The reference uses Open3D segment_plane on an unseeded
 * np.random.choice subsample (third-party, random); this is the build's deterministic seeded
 * definition (DESIGN.md "RANSAC"): z < -1.5 candidates, systematic subsample to 5000, all pixels if
 * fewer than 800, 100 hypotheses of 10 points, 0.1 m inlier distance, refit on the inliers.
 *   frame_ids dev i64 [B] or NULL a stable identity per frame (its datalist index): frame b draws with
 *                                seed + (uint32)frame_ids[b] (NULL: seed + b), so a file's plane -- and its
 *                                bitstream -- does not depend on the batch it travels in
 *   ground   dev f64 [B,4] out   plane a,b,c,d (unit normal) per frame
 *   inliers  dev i32 [B]   out   inlier count of the winning hypothesis (may be NULL)            */
int rpcc_ground_ransac(const float *ri, const float *tm, int B, int P, uint32_t seed, const int64_t *frame_ids,
                       double *ground, int32_t *inliers, void *stream);

/* ---- a3+a5: back-projection + vertical ground residual + FPS state init -------------------- *
 * replaces PCTransformer.range_image_to_point_cloud (dataset/transformer.py:94-101) and
 * PointCloudSegment.calc_plane_residual_vertical cpu branch + mask (utils/segment_utils.py:44-47,
 * 119-120).  Candidates (depth_dif > threshold) get temp = 1e10 (ops/fps/fps_utils.py:26), all other
 * pixels temp = -1 (never selectable).
 *   ri       dev f32 [B,P]
 *   tm       dev f32 [P,3]       transform_map (dataset/transformer.py:41-54)
 *   ground   dev f64 [B,4]       plane a,b,c,d per frame
 *   temp     dev f32 [B,P]  out
 *   info     dev i32 [B,8]  out  {n_left, first candidate pixel (P if none), nnz, fps_table valid,
 *                                 first EMPTY pixel that is a candidate (P if none), 3 spare}       */
int rpcc_ground_mask(const float *ri, const float *tm, const double *ground, double threshold, int B, int H, int W,
                     float *temp, int32_t *info, void *fps_table, void *stream);
/* fps_table (optional, rpcc_fps_table_bytes(B,H,W) bytes, or NULL): when given, the kernel also runs the
 * first pass of the farthest point sampling (distance of every candidate to the first centre, per-tile
 * bounding boxes and maxima) and temp holds min(1e10, that distance) for frames with info[b][3] == 1;
 * pass the same table to rpcc_fps_range.  Results are identical with and without it. */
size_t rpcc_fps_table_bytes(int B, int H, int W);

/* ---- a6: farthest point sampling ----------------------------------------------------------- *
 * rpcc_fps_xyz replaces furthest_point_sampling_wrapper(b,n,m,points,temp,idx)
 * (ops/fps/src/sampling.cpp:24-37, kernel ops/fps/src/sampling_gpu.cu:24-140): same arguments, same
 * caller-initialised temp (1e10), idx[b][0] = 0, strict '>' arg-max, ties -> lowest index.
 *   points dev f32 [B,N,3], temp dev f32 [B,N] in/out, idx dev i32 [B,M] out                      */
int rpcc_fps_xyz(int B, int N, int M, const float *points, float *temp, int32_t *idx, void *stream);

/* rpcc_fps_range is the same sampling run directly on the range image (xyz = ri * tm recomputed in
 * registers; pixels with temp < 0 are not candidates).  It selects the pixels the reference selects
 * on the compacted candidate list (utils/segment_utils.py:120-124).
 *   temp     dev f32 [B,P] in/out from rpcc_ground_mask (candidates share one initial value, others < 0); on return
 *                                the running minimum distance of every candidate, as the reference kernel leaves it
 *   info     dev i32 [B,8]       from rpcc_ground_mask (first candidate = start point; first empty candidate)
 *   cen_pix  dev i32 [B,M]  out  pixel index of each centre
 *   centers  dev f32 [B,M,3] out cluster_centers
 *   flags    0, or RPCC_FPS_BRUTEFORCE for the one-pass-per-centre kernel (identical results), or the CUDA-binary modes
 *            RPCC_FPS_FMA1 / RPCC_FPS_FMA2 / RPCC_FPS_TIE_CUDA (fps_table must be NULL then; k of the tie rule = rank of
 *            the pixel among the candidates, block = opt_n_threads(n_left))                              */
int rpcc_fps_range(const float *ri, const float *tm, float *temp, const int32_t *info, int B, int H, int W, int M,
                   int32_t *cen_pix, float *centers, int flags,
                   const void *fps_table /* from rpcc_ground_mask, or NULL */, void *stream);

/* The brute-force form of rpcc_fps_xyz (one full pass per centre): the test reference of the tile-pruned kernels. */
int rpcc_fps_xyz_bruteforce(int B, int N, int M, const float *points, float *temp, int32_t *idx, void *stream);
/* rpcc_fps_xyz with flags: RPCC_FPS_BRUTEFORCE, RPCC_FPS_FMA1 / RPCC_FPS_FMA2, RPCC_FPS_TIE_CUDA (k = the point's index,
 * block = opt_n_threads(N) as sampling_gpu.cu:9-13 computes it). */
int rpcc_fps_xyz_mode(int B, int N, int M, const float *points, float *temp, int32_t *idx, int flags, void *stream);
/* rpcc_fps_xyz sends every list to the kernel its point order suits: tiles of 256 consecutive points that are compact in space (the reference's
 * row-major candidate list, a sweep in its stored order) -> the tile-pruned kernel; no locality (a shuffled cloud) -> one pass per centre.  Same
 * indices either way.  Test hook: the decision alone.   marks dev i32 [B] out: -1 = one pass per centre, 0 = tile-pruned */
int rpcc_fps_xyz_probe(int B, int N, const float *points, int32_t *marks, void *stream);

/* ---- a7: ground / cluster assignment + relabel ---------------------------------------------- *
 * replaces calc_plane_residual_depth, calc_cluster_residual_radius, concatenate + argmax and the
 * relabel (utils/segment_utils.py:21-23,64-67,127-131,168-169).
 *   seg      dev u8 [B,P] out labels                                                              */
int rpcc_assign(const float *ri, const float *tm, const double *ground, const float *centers, int B, int H, int W,
                int M, uint8_t *seg, void *stream);

/* ---- a8: point model ------------------------------------------------------------------------- *
 * replaces segment_utils_cpp.point_modeling (cpp_modules.cpp:471-518) + the model_param assembly
 * (utils/segment_utils.py:183-185, tools/compress.py:102), then the pybind11 fp64->fp32 cast.
 *   model    dev f32 [B,K,4] out row 0 = (float)ground, row 1 = 0, row k = [0,0,0,mean_k]
 *                                (NaN 0xFFC00000 for a label without pixels)
 *   counts   dev i32 [B,K]   out pixels per label
 *   ws       dev, rpcc_workspace_bytes                                                            */
int rpcc_point_model(const float *ri, const uint8_t *seg, const double *ground, int B, int P, int M, float *model,
                     int32_t *counts, void *ws, void *stream);

/* ---- a9: plane model --------------------------------------------------------------------------- *
 * replaces cluster_modeling('plane') (utils/segment_utils.py:188-216) incl. plane_angle_validation
 * (:84-93) and the fp32 numpy-mean fallbacks.  RANSAC (ransac_n = 4, 10 iterations, 0.1 m) is the
 * build's seeded specification (Open3D in the reference); label k of frame b uses hash(seed, id_b, k) with
 * id_b = frame_ids[b] (dev i64 [B]) or b when frame_ids is NULL.
 *   inject_planes dev f64 [B,K,4] or NULL: test hook -- row (b,k) replaces the RANSAC result of label k (what the
 *            fixtures of tests/golden/pins_*.npz do to the reference through ransac_plane_segmentation), so that the
 *            angle validation and the mean fallbacks can be compared with the genuine cluster_modeling('plane')
 *   ground   dev f64 [B,4] or NULL   copied (as fp32) into row 0
 *   cos_cut  HOST double: a plane is rejected when some pixel has |n.t|/|n|*|t| <= cos_cut, i.e.
 *            arccos(.) > angle threshold; the caller derives it from its own arccos (ops.plane_model)
 *   model    dev f32 [B,K,4] out;  counts dev i32 [B,K] out;  ws rpcc_plane_workspace_bytes(B,P,M)   */
size_t rpcc_plane_workspace_bytes(int B, int P, int M);
int rpcc_plane_model(const float *ri, const float *tm, const uint8_t *seg, const double *ground, int B, int P, int M,
                     double cos_cut, uint32_t seed, const int64_t *frame_ids, const double *inject_planes, float *model,
                     int32_t *counts, void *ws, void *stream);

/* ---- a10+a11(+a13): intra-prediction + residual + quantisation + ordered scatter ------------- *
 * replaces segment_utils_cpp.intra_predict (cpp_modules.cpp:248-285), residual = ri - pred
 * (tools/compress.py:106), quantization_utils_cpp.uniform_quantize (cpp_modules.cpp:288-334) and the
 * quantising half of nonuniform_quantize (cpp_modules.cpp:404-422).
 *   acc         quantisation step (= 2*accuracy, tools/compress.py:46) as C float (uniform)
 *   label_acc   dev f32 [B,K] per-label step from rpcc_salience (non-uniform), or NULL
 *   residual_in dev f32 [B,P] residual supplied by the caller (QuantizationModule.quantize_residual's
 *               own argument, utils/compress_utils.py:57), or NULL to compute ri - pred here.  With
 *               residual_in given and pred == NULL nothing is predicted: ri, tm and model may be NULL
 *               (exactly uniform_quantize(seg_idx, residual, acc), cpp_modules.cpp:288)
 *   q16         dev i16 [B,P] out   per frame: nnz values grouped by label ascending, row-major inside
 *                                   a label, already cast to int16 (utils/compress_utils.py:142)
 *   q32         dev i32 [B,P] out   same as int32 (what the quantisers return); either may be NULL
 *   nnz         dev i32 [B]   out
 *   pred        dev f32 [B,P] out   optional (NULL to skip)                                          */
int rpcc_predict_quantize(const float *ri, const float *tm, const uint8_t *seg, const float *model,
                          const float *label_acc, const float *residual_in, float acc, int B, int P, int M,
                          int16_t *q16, int32_t *q32, int32_t *nnz, float *pred, void *ws, void *stream);

/* a10 alone: segment_utils_cpp.intra_predict (cpp_modules.cpp:248-285) -> pred dev f32 [B,P]. */
int rpcc_intra_predict(const uint8_t *seg, const float *model, const float *tm, int B, int P, int M, float *pred,
                       void *stream);

/* ---- a12: key points ---------------------------------------------------------------------------- *
 * replaces feature_extractor_cpp.extract_features_with_segment (cpp_modules.cpp:28-121, mark_as_picked
 * :10-25) with zero-initialised outputs (the reference leaves unwritten cells uninitialised).
 *   feat          dev f32 [B,H,W] out   curvature feature
 *   key_point_map dev u8  [B,H,W] out   0 none, 1 flat, 2 less sharp, 3 sharp
 * Limits: W <= 4096 (a row lives in one workgroup's registers / LDS), 1 <= feature_region <= 16, segments >= 1;
 * any chunk length (W / segments) and any sharp / less_sharp / flat counts.                            */
int rpcc_extract_features(const float *ri, const uint8_t *seg, int B, int H, int W, int feature_region, int segments,
                          int sharp_num, int less_sharp_num, int flat_num, float *feat, uint8_t *key_point_map,
                          void *stream);

/* ---- a13: salience levels ------------------------------------------------------------------------ *
 * replaces the per-label level selection of nonuniform_quantize (cpp_modules.cpp:355-405).
 *   level_kp_num HOST i32 [levels], level_acc HOST f32 [levels]  (base step + delta, compress_utils.py:48)
 *   salience   dev u8  [B,K] out   level per label;  label_acc dev f32 [B,K] out  its step            */
int rpcc_salience(const uint8_t *seg, const uint8_t *key_point_map, const int32_t *level_kp_num, const float *level_acc,
                  int levels, int ground_level, int B, int P, int M, uint8_t *salience, float *label_acc, void *stream);

/* ---- a3: back-projection as its own entry --------------------------------------------------- *
 * replaces PCTransformer.range_image_to_point_cloud (dataset/transformer.py:94-101).
 *   pc       dev f32 [B,P,3] out  ri[...,None] * transform_map                                     */
int rpcc_backproject(const float *ri, const float *tm, int B, int P, float *pc, void *stream);

/* ---- f1: contour map + index sequence --------------------------------------------------------- *
 * replaces contour_utils_cpp.extract_contour (cpp_modules.cpp:521-558) and the casts/packing of
 * compress_point_cloud (utils/compress_utils.py:156-160).
 *   contour_bits dev u8  [B, ceil(P/8)] out  np.packbits(contour_map) (first pixel = MSB)
 *   idx_sequence dev u16 [B,P]          out  labels at the contour positions, row-major; nseq[b] entries
 *   ws           dev, rpcc_codec_workspace_bytes(B,P,M)                                            */
size_t rpcc_codec_workspace_bytes(int B, int P, int M);
int rpcc_contour_encode(const uint8_t *seg, int B, int H, int W, uint8_t *contour_bits, uint16_t *idx_sequence,
                        int32_t *nseq, void *ws, void *stream);

/* ---- f3: decoder ------------------------------------------------------------------------------- *
 * rpcc_contour_decode replaces np.unpackbits + contour_utils_cpp.recover_map (cpp_modules.cpp:561-593,
 * utils/compress_utils.py:202-206).
 * rpcc_decode replaces QuantizationModule.dequantize_residual (utils/compress_utils.py:114-132),
 * intra_predict, range_image_rec = pred + residual and range_image_to_point_cloud
 * (tools/decompress.py:88-112).
 *   q16        dev i16 [B,P]   label-ordered quantised residuals (as stored in the bitstream)
 *   model      dev f32 [B,K,4] plane_param from the bitstream
 *   level_acc  HOST f64 [max(levels,1)]  quantisation step(s): uniform -> levels = 0 and level_acc[0]
 *   salience   dev u8 [B,K]    per-label level (non-uniform only)
 *   ri_rec     dev f32 [B,P] out ;  pc_rec dev f32 [B,P,3] out (may be NULL)                        */
int rpcc_contour_decode(const uint8_t *contour_bits, const uint16_t *idx_sequence, int B, int H, int W, uint8_t *seg,
                        void *ws, void *stream);
int rpcc_decode(const uint8_t *seg, const int16_t *q16, const float *model, const float *tm, const double *level_acc,
                int levels, const uint8_t *salience, int B, int P, int M, float *ri_rec, float *pc_rec, void *ws,
                void *stream);

/* ---- f2: the batch's residual stream, frames back to back ------------------------------------ *
 * replaces the per-frame `residual_quantized` arrays of compress_point_cloud (utils/compress_utils.py:142,160)
 * for a batch: packed = concat_b q16[b][:nnz[b]] (prefix sums taken on the device; no host round trip).
 *   q16      dev i16 [B,P]   label-ordered residuals (rpcc_predict_quantize / rpcc_compress_batch)
 *   nnz      dev i32 [B]     entries per frame
 *   packed   dev i16 [capacity] out   entries past `capacity` are dropped (capacity >= sum of the frames' input
 *                                     points is always enough: a pixel holds at least one point)
 *   total    dev i64 [1] out (may be NULL)  sum of nnz = entries written
 * What a rank hands to the D2H copy or to the RCCL gather of payloads instead of the padded array. */
int rpcc_pack_payload(const int16_t *q16, const int32_t *nnz, int B, int P, int16_t *packed, int64_t capacity,
                      int64_t *total, void *stream);

/* ---- fused batch entry: a2..a13 for B frames (FPS segmentation; uniform / non-uniform framework; point / plane model) *
 * The batched counterpart of the body of tools/compress.py:93-125 /
 * tools/compress_datalist.py:91-125 with the ground model supplied by the caller or fitted inside. */
/* QuantizationModule's non-uniform settings (utils/compress_utils.py:36-54, cfgs/compressor.yaml:26-36) */
typedef struct rpcc_nonuniform_cfg {
    int32_t levels;              /* 1..8 */
    int32_t level_kp_num[8];     /* level_key_point_num */
    float level_acc[8];          /* base step + level_delta_acc, as C float (utils/compress_utils.py:48) */
    int32_t ground_level;        /* ground_salience_level */
    int32_t feature_region, segments, sharp_num, less_sharp_num, flat_num;
} rpcc_nonuniform_cfg;

typedef struct rpcc_batch_io {
    const float *xyz;        /* dev f32 [total,3] (or [total,4] rows with point_stride_bytes = 16) */
    const int64_t *offsets;  /* dev i64 [B+1] */
    int64_t total;
    const float *tm;         /* dev f32 [P,3] */
    double *ground;          /* dev f64 [B,4]  in (ground_seed < 0: injected models) / out (fitted here) */
    int64_t ground_seed;     /* >= 0: fit the ground plane with rpcc_ground_ransac(seed = ground_seed, frame_ids) */
    const int64_t *frame_ids;/* dev i64 [B] or NULL: stable identity of every frame (seeds; see rpcc_ground_ransac) */
    float *ri;               /* dev f32 [B,P] out */
    uint8_t *seg;            /* dev u8  [B,P] out */
    int32_t *cen_pix;        /* dev i32 [B,M] out */
    float *centers;          /* dev f32 [B,M,3] out */
    float *model;            /* dev f32 [B,K,4] out */
    int32_t *counts;         /* dev i32 [B,K] out */
    int16_t *q16;            /* dev i16 [B,P] out */
    int32_t *nnz;            /* dev i32 [B] out */
    int32_t *info;           /* dev i32 [B,8] out */
    int32_t flags;           /* 0, RPCC_FPS_BRUTEFORCE, or the CUDA-binary FPS modes RPCC_FPS_FMA1 / _FMA2 / _TIE_CUDA; | RPCC_PROJECT_ORDER_PROBE /
                                RPCC_PROJECT_FORCE_ORDERED (see rpcc_project_ordered) */
    void *timer;             /* rpcc_timer_create() handle or NULL: times this call's FPS launch (bench.py) */
    /* framework / model selection (tools/compress.py:109-124, cfgs/compressor.yaml: compress_framework, modeling_method) */
    int32_t model_method;    /* 0: point model (a8);  1: plane model (a9: rpcc_plane_model with plane_cos_cut, plane_seed,
                                frame_ids); ws must then hold rpcc_workspace_bytes_general() bytes */
    double plane_cos_cut;    /* see rpcc_plane_model */
    int64_t plane_seed;
    const rpcc_nonuniform_cfg *nonuniform; /* HOST, NULL: uniform framework (step = acc); else key points + salience
                                levels + per-label steps (a12, a13), ws of rpcc_workspace_bytes_general() bytes */
    uint8_t *salience;       /* dev u8 [B,K] out (non-uniform) */
    uint8_t *key_point_map;  /* dev u8 [B,P] out (non-uniform) */
    int32_t point_stride_bytes; /* bytes per point of `xyz`: 0 or 12 = packed xyz; 16 = (x, y, z, intensity) rows as stored in a
                                KITTI .bin (see rpcc_project_strided) */
} rpcc_batch_io;

size_t rpcc_workspace_bytes(int B, int P, int M, int64_t total_points);
size_t rpcc_workspace_bytes_general(int B, int P, int M, int64_t total_points); /* plane model and / or non-uniform framework */
int rpcc_compress_batch(const rpcc_batch_io *io, int B, rpcc_geom g, int M, double ground_threshold, float acc,
                        void *ws, void *stream);

/* A subset of rpcc_compress_batch's stages, in order: bit 0 projection, 1 ground fit, 2 mask (+ first FPS pass), 3 FPS, 4 assignment + label histograms +
 * scan (+ the plane model's label order), 5 plane fits, 6 key points + salience + quantiser.  For callers that interleave the stages of several batches on
 * several streams themselves; every stage needs the stages before it to have run on the same io / ws.  All bits (127) = rpcc_compress_batch. */
#define RPCC_STAGE_ALL 127
int rpcc_compress_batch_stages(const rpcc_batch_io *io, int B, rpcc_geom g, int M, double ground_threshold, float acc, void *ws,
                               int stage_mask, void *stream);

/* The same for a batch that holds sweeps of several lidar geometries (variable H x W inside one call: BASELINE configs[4]; the
 * reference compresses such a list frame by frame with one dataset / transformer per lidar, tools/compress_datalist.py:160-206,
 * the YAML files of dataset/lidar_cfg).  The frames are grouped by geometry: group i = ios[i] (its own buffers, laid out for Bs[i] frames of
 * geoms[i], exactly as for rpcc_compress_batch), workspace wss[i] of rpcc_workspace_bytes[_general](Bs[i], P_i, M, total_i) bytes.
 * Everything is queued on `stream`; the kernels with one workgroup per frame or per label (ground RANSAC, FPS, plane fits) run
 * as ONE launch over the frames of all groups, the pixel-parallel kernels group after group.  Results per group are what
 * rpcc_compress_batch returns for that group alone.  G <= RPCC_MAX_GROUPS; ground_seed / model_method / nonuniform may differ
 * between the groups. */
#define RPCC_MAX_GROUPS 4
int rpcc_compress_batch_mixed(const rpcc_batch_io *ios, const int *Bs, const rpcc_geom *geoms, int G, int M,
                              double ground_threshold, float acc, void *const *wss, void *stream);

/* cluster_num above RPCC_MAX_CLUSTERS (the reference takes any value, cfgs/compressor.yaml:22; its labels travel as uint16,
 * utils/compress_utils.py:160): the same batch with `io->seg` pointing to uint16 [B,P] labels, 255 <= M <= RPCC_MAX_CLUSTERS_WIDE (smaller M work too).
 * Same stages, same arithmetic, same results as rpcc_compress_batch -- one thread per pixel or label, per-label totals by global atomics, the ordered
 * scatter through one stable radix sort of (frame, label) keys: written for correctness, not for speed (csrc/wide_kernels.h).  All four framework / model
 * combinations; the CUDA-binary FPS modes are not available here.  ws: rpcc_wide_workspace_bytes(B, P, M, total_points) bytes.
 * The container's side of it: the contour codec and the decoder body on uint16 label maps (rpcc_decode_wide: ws of rpcc_wide_workspace_bytes(B, P, M, 0)).
 * rpcc_decode_wide takes label maps of foreign streams: a label above M + 1 (no row of `model`) is read as M + 1 -- memory-safe, the pixel's value is then
 * meaningless as the stream's was; callers that must reject such a stream check the label range first (compress_utils.decode_frame does). */
#define RPCC_MAX_CLUSTERS_WIDE 65533
/* Up to RPCC_MAX_CLUSTERS_MID clusters the per-label tables of the tuned kernels still fit LDS: rpcc_compress_batch_wide then runs the assignment, the
 * label histogram, the plane model's label order and the quantiser of rpcc_compress_batch on uint16 labels (250 k frames/s at 300 clusters against
 * 89 k through the radix sort), and the reference's STAGE seams exist for such counts too -- the uint16 forms of rpcc_assign, rpcc_point_model, rpcc_intra_predict
 * and rpcc_predict_quantize (same arguments, `seg` as uint16; ws of rpcc_workspace_bytes(B, P, M, 0) bytes):
 * PointCloudSegment.segment / cluster_modeling('point') / intra_predict and the uniform quantize_residual stage by stage at cluster_num 255 .. 1022. */
#define RPCC_MAX_CLUSTERS_MID 1022
int rpcc_assign_wide(const float *ri, const float *tm, const double *ground, const float *centers, int B, int H, int W, int M, uint16_t *seg, void *stream);
int rpcc_point_model_wide(const float *ri, const uint16_t *seg, const double *ground, int B, int P, int M, float *model, int32_t *counts, void *ws, void *stream);
int rpcc_plane_model_wide(const float *ri, const float *tm, const uint16_t *seg, const double *ground, int B, int P, int M, double cos_cut, uint32_t seed,
                          const int64_t *frame_ids, const double *inject_planes, float *model, int32_t *counts, void *ws, void *stream);
int rpcc_intra_predict_wide(const uint16_t *seg, const float *model, const float *tm, int B, int P, int M, float *pred, void *stream);
int rpcc_extract_features_wide(const float *ri, const uint16_t *seg, int B, int H, int W, int feature_region, int segments, int sharp_num,
                               int less_sharp_num, int flat_num, float *feat, uint8_t *key_point_map, void *stream);
int rpcc_salience_wide(const uint16_t *seg, const uint8_t *key_point_map, const int32_t *level_kp_num, const float *level_acc, int levels,
                       int ground_level, int B, int P, int M, uint8_t *salience, float *label_acc, void *stream);
int rpcc_predict_quantize_wide(const float *ri, const float *tm, const uint16_t *seg, const float *model, const float *label_acc, const float *residual_in,
                               float acc, int B, int P, int M, int16_t *q16, int32_t *q32, int32_t *nnz, float *pred, void *ws, void *stream);
size_t rpcc_wide_workspace_bytes(int B, int P, int M, int64_t total_points);
int rpcc_compress_batch_wide(const rpcc_batch_io *io, int B, rpcc_geom g, int M, double ground_threshold, float acc, void *ws, void *stream);
int rpcc_contour_encode_wide(const uint16_t *seg, int B, int H, int W, uint8_t *contour_bits, uint16_t *idx_sequence, int32_t *nseq, void *ws, void *stream);
int rpcc_contour_decode_wide(const uint8_t *contour_bits, const uint16_t *idx_sequence, int B, int H, int W, uint16_t *seg, void *ws, void *stream);
int rpcc_decode_wide(const uint16_t *seg, const int16_t *q16, const float *model, const float *tm, const double *level_acc, int levels,
                     const uint8_t *salience, int B, int P, int M, float *ri_rec, float *pc_rec, void *ws, void *stream);

/* Developer hook (libraries built with -DRPCC_DEVTRACE only; the shipped one returns RPCC_ERR_ARG for a non-NULL buffer):
 * register a device int64 buffer of at least RPCC_DEBUG_STAMPS_WORDS words; instrumented kernels store the shader clock at
 * phase boundaries, the FPS kernel its per-phase cycle sums and per-iteration tile counts.  NULL disables it (default). */
#define RPCC_DEBUG_STAMPS_WORDS (4096 + 16 * 128 * 8)
int rpcc_debug_stamps(void *dev_i64_buffer);

/* Timer objects for bench.py: a handle given in rpcc_batch_io.timer makes the call record hipEvents around its FPS
 * launch on the call's stream; rpcc_timer_read returns the accumulated milliseconds and the launch count since the
 * last read (synchronises the events).  One handle per measuring thread; the library keeps no global timing state.
 * rpcc_timer_reserve creates the events of the next `launches` timed launches ahead of time (event creation is not
 * cheap and would otherwise fall into the caller's timed region). */
void *rpcc_timer_create(void);
void rpcc_timer_destroy(void *timer);
int rpcc_timer_reserve(void *timer, int launches);
int rpcc_timer_read(void *timer, double *ms, int *launches);

#ifdef __cplusplus
}
#endif
#endif
