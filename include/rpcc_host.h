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

int rpcc_host_version(void);   /* 101 */

#define RPCC_HOST_ERR_ARG (-1)

/* src, src_bytes: [nframes * narrays] arrays in container order (salience_level first for the non-uniform framework, then
 * contour_map, idx_sequence, plane_param, residual_quantized); frame f's container is written at dst + dst_off[f] and may use
 * dst_off[f + 1] - dst_off[f] bytes (dst_off: [nframes + 1]); dst_bytes: [nframes] container lengths.
 * Returns 0; RPCC_HOST_ERR_ARG for a bad argument; -(16 + frame) when that frame does not fit its region or libbz2 reports an
 * error. */
int rpcc_host_pack_bz2(int nframes, int narrays, const void *const *src, const uint32_t *src_bytes, uint8_t *dst,
                       const uint64_t *dst_off, uint32_t *dst_bytes);

#ifdef __cplusplus
}
#endif
#endif
