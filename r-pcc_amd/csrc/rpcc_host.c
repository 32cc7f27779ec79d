// rpcc_host.c -- host side of f2: the .rpcc container of MANY frames in one call.
//
// The reference packs a frame as [int32 length | entropy-coded bytes] per array (utils/compress_utils.py:167-179) and
// entropy-codes every array with Python's bz2.compress (BasicCompressor, :199-214), one frame per pool thread
// (tools/compress_datalist.py:202-206).  From Python that is four or five interpreter round trips per frame; with more than
// ~32 pool threads the hand-over of the interpreter lock between them, not bzip2, bounds the rate (measured: 880 frames/s on
// 32 threads, 730 on 128).  Here a pool thread enters the library once per chunk of frames and stays outside the interpreter
// until the chunk's containers are complete.  The bytes are the ones bz2.compress produces: the same libbz2, block size
// 9, default work factor (BZ2_bzBuffToBuffCompress is BZ2_bzCompressInit + one BZ_FINISH call; tests/test_host_pack.py).
#include <stddef.h>
#include <stdint.h>
#include <string.h>

// libbz2's one-shot interface (bzlib.h is not installed in the image; the library the interpreter's bz2 module links is)
extern int BZ2_bzBuffToBuffCompress(char *dest, unsigned int *destLen, char *source, unsigned int sourceLen,
                                    int blockSize100k, int verbosity, int workFactor);

int rpcc_host_version(void) { return 100; }

// src / src_bytes: [nframes * narrays] arrays in container order; dst: nframes regions of dst_stride bytes; dst_bytes
// [nframes] container lengths out.  Returns 0, or -(1 + frame) when a frame does not fit its region or libbz2 fails.
int rpcc_host_pack_bz2(int nframes, int narrays, const void *const *src, const uint32_t *src_bytes, uint8_t *dst,
                       size_t dst_stride, uint32_t *dst_bytes) {
    if (nframes < 0 || narrays <= 0 || !src || !src_bytes || !dst || !dst_bytes) return -1;
    for (int f = 0; f < nframes; f++) {
        uint8_t *out = dst + (size_t)f * dst_stride;
        size_t off = 0;
        for (int a = 0; a < narrays; a++) {
            const size_t i = (size_t)f * narrays + a;
            if (off + 4 > dst_stride) return -(1 + f);
            const size_t room = dst_stride - off - 4;
            unsigned int n = room > 0xFFFFFFFFu ? 0xFFFFFFFFu : (unsigned int)room;
            static char none;
            char *s = src[i] ? (char *)(uintptr_t)src[i] : &none;   // an empty array may come with a null pointer
            if (BZ2_bzBuffToBuffCompress((char *)out + off + 4, &n, s, src_bytes[i], 9, 0, 0) != 0) return -(1 + f);
            const int32_t len = (int32_t)n;
            memcpy(out + off, &len, 4);   // struct.pack("i", len)
            off += 4 + n;
        }
        dst_bytes[f] = (uint32_t)off;
    }
    return 0;
}
