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

int rpcc_host_version(void) { return 101; }

// src / src_bytes: [nframes * narrays] arrays in container order; frame f's container is written at dst + dst_off[f] and may
// use dst_off[f + 1] - dst_off[f] bytes (dst_off: [nframes + 1], so every frame gets the room ITS arrays need instead of
// the largest frame's); dst_bytes [nframes] container lengths out.
// Returns 0; RPCC_HOST_ERR_ARG (-1) for a bad argument; -(16 + frame) when that frame does not fit its region or libbz2
// reports an error.
#define RPCC_HOST_ERR_ARG (-1)
int rpcc_host_pack_bz2(int nframes, int narrays, const void *const *src, const uint32_t *src_bytes, uint8_t *dst,
                       const uint64_t *dst_off, uint32_t *dst_bytes) {
    if (nframes < 0 || narrays <= 0 || !src || !src_bytes || !dst || !dst_off || !dst_bytes) return RPCC_HOST_ERR_ARG;
    for (int f = 0; f < nframes; f++) {
        if (dst_off[f + 1] < dst_off[f]) return RPCC_HOST_ERR_ARG;
        uint8_t *out = dst + dst_off[f];
        const size_t region = (size_t)(dst_off[f + 1] - dst_off[f]);
        size_t off = 0;
        for (int a = 0; a < narrays; a++) {
            const size_t i = (size_t)f * narrays + a;
            if (off + 4 > region) return -(16 + f);
            const size_t room = region - off - 4;
            unsigned int n = room > 0xFFFFFFFFu ? 0xFFFFFFFFu : (unsigned int)room;
            static char none;
            char *s = src[i] ? (char *)(uintptr_t)src[i] : &none;   // an empty array may come with a null pointer
            if (BZ2_bzBuffToBuffCompress((char *)out + off + 4, &n, s, src_bytes[i], 9, 0, 0) != 0) return -(16 + f);
            const int32_t len = (int32_t)n;
            memcpy(out + off, &len, 4);   // struct.pack("i", len)
            off += 4 + n;
        }
        dst_bytes[f] = (uint32_t)off;
    }
    return 0;
}
