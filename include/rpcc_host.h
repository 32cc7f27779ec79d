/* rpcc_host.h -- host-side helper of the f2 row (container + entropy coder), plain C, no GPU.
 *
 * Replaces, for a chunk of frames at once, what the reference does per frame in Python:
 *   BasicCompressor.compress_dict (utils/compress_utils.py:199-214, bz2.compress per array) followed by
 *   save_compressed_bitstream's record layout [int32 length | bytes] (utils/compress_utils.py:167-179),
 * called from one pool thread per frame in tools/compress_datalist.py:202-206.  Same bytes (same libbz2, level 9).
 * Built by r-pcc_amd/build.py into r-pcc_amd/lib/librpcc_host.so; used by r-pcc_amd/compress_utils.py (pack_frames).
 */
#ifndef RPCC_HOST_H
#define RPCC_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int rpcc_host_version(void);

/* src, src_bytes: [nframes * narrays] arrays in container order (salience_level first for the non-uniform framework, then
 * contour_map, idx_sequence, plane_param, residual_quantized); dst: nframes regions of dst_stride bytes each; dst_bytes:
 * [nframes] container lengths.  Returns 0, or -(1 + frame) when a frame does not fit its region or libbz2 reports an error. */
int rpcc_host_pack_bz2(int nframes, int narrays, const void *const *src, const uint32_t *src_bytes, uint8_t *dst,
                       size_t dst_stride, uint32_t *dst_bytes);

#ifdef __cplusplus
}
#endif
#endif
