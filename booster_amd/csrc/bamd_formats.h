// bamd_formats.h — GGUF K-quant block formats (as they sit in the file) and the MI355X "wave-stream"
// layout they are repacked into at load time.
//
// File formats follow the reference's block definitions (cpp/ggml/src/ggml-common.h:267-316):
//   Q4_K 144 B = {f16 d, f16 dmin, u8 scales[12], u8 qs[128]}
//   Q5_K 176 B = {f16 d, f16 dmin, u8 scales[12], u8 qh[32], u8 qs[128]}
//   Q6_K 210 B = {u8 ql[128], u8 qh[64], i8 scales[16], f16 d}
//   Q8_K 292 B = {f32 d, i8 qs[256], i16 bsums[16]}   (activations only; never stored in HBM here)
//
// Wave-stream layout (our design; the GGUF file itself is untouched).  A matrix [nrows][K] is cut into
// row-groups of 8 rows; for each row-group the K/256 super-blocks follow one another as RECORDS, and a record
// holds the same super-block index of all 8 rows, arranged so that wave lane (r*8 + e) — row r of the group,
// SIMD-lane e of the reference's 256-bit AVX2 registers — finds its bytes at lane*16:
//
//   Q4_K record 1152 B: [qs   : lane*16 -> 4 dwords j=0..3 = file qs[32j+4e .. +3]      ] 1024 B
//                       [hdr  : r*16    -> file bytes 0..15 (d, dmin, scales[12])       ]  128 B
//   Q5_K record 1408 B: [qs 1024 B as Q4_K][qh: lane*4 -> file qh[4e..4e+3] 256 B][hdr 128 B]
//   (BAMD_XSCALES = 1, an experiment kept behind the macro: hdr = {d | dmin<<16, sc[0..3], sc[4..7], mn[0..3]} + 32 B of mn[4..7])
//   Q6_K record 1680 B: [ql   : lane*16 -> 4 dwords j=0..3 = file ql[32j+4e .. +3]      ] 1024 B
//                       [qh   : lane*8  -> 2 dwords m=0,1 = file qh[32m+4e .. +3]      ]  512 B
//                       [sc   : r*16    -> byte hi*8+c = file scales[2c+hi]             ]  128 B
//                       [d    : r*2     -> f16 d                                        ]   16 B
//
// Record bytes = 8 x file block bytes, so HBM traffic per weight is exactly the GGUF's bits per weight, every
// wave load instruction is one contiguous, 16-byte-per-lane kilobyte, and consecutive records of a row-group
// are consecutive in memory (a pure sequential stream per wave).
#pragma once
#include <stdint.h>
#include <stddef.h>

#define BAMD_QK_K 256
#ifdef __HIPCC__
#define BAMD_HD __host__ __device__
#else
#define BAMD_HD
#endif

enum bamd_type { BAMD_F32 = 0, BAMD_F16 = 1, BAMD_Q4_K = 12, BAMD_Q5_K = 13, BAMD_Q6_K = 14 };

BAMD_HD static inline int bamd_block_bytes(int t) { return t == BAMD_Q4_K ? 144 : t == BAMD_Q5_K ? 176 : t == BAMD_Q6_K ? 210 : 0; }
#ifndef BAMD_XSCALES
#define BAMD_XSCALES 0          /* 0: the file's 12 packed scale bytes per row (records of 1152 / 1408 B = 8 x the GGUF block); 1: unpacked scales and
                                   mins, a byte each (1184 / 1440 B) — measured SLOWER on the MI355X in round 2 (gate/up 14.8 vs 13.2 us, decode 649 vs
                                   666 tok/s): 9 vector instructions saved per record do not pay for +2.8 % bytes, a third request per record and
                                   records that no longer start on a 128-byte line */
#endif
#define BAMD_RECB_Q4K (BAMD_XSCALES ? 1184 : 1152)
#define BAMD_RECB_Q5K (BAMD_XSCALES ? 1440 : 1408)
BAMD_HD static inline int bamd_record_bytes(int t) { return t == BAMD_Q4_K ? BAMD_RECB_Q4K : t == BAMD_Q5_K ? BAMD_RECB_Q5K : t == BAMD_Q6_K ? 1680 : 0; }   // wave-stream record: 8 rows x 1 super-block
// bytes of the wave-stream copy of a K-quant matrix [nrows_pad (multiple of 8)][K]
BAMD_HD static inline size_t bamd_stream_bytes(int t, int64_t k, int64_t nrows_pad) { return (size_t) (nrows_pad / 8) * (size_t) (k / BAMD_QK_K) * (size_t) bamd_record_bytes(t); }
BAMD_HD static inline int bamd_is_kquant(int t) { return t == BAMD_Q4_K || t == BAMD_Q5_K || t == BAMD_Q6_K; }
BAMD_HD static inline size_t bamd_row_bytes(int t, int64_t k) {
    return t == BAMD_F32 ? (size_t) k * 4 : t == BAMD_F16 ? (size_t) k * 2 : (size_t) (k / BAMD_QK_K) * bamd_block_bytes(t);
}
