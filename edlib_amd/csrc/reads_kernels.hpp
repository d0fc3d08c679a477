// reads_kernels.hpp -- launch interface of the "reads-per-lane" scan family
// (many short queries against one shared target; BASELINE.json configs 2/3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace edlib_amd {

constexpr int kMaxReadWords = 8;      // queries up to 256 symbols: one kernel instance per word count, every mode
// HW, banded kernel only: queries of 257..320 / ..384 / ..448 / ..512 symbols run in groups of 10 / 12 / 14 / 16 words (targets
// of up to 8 symbols), 513..768 / 769..1024 in groups of 24 / 32 words (four symbols); the row m-1 of a lane may sit in any of
// the group's last four (eight) words
constexpr int kMaxLongReadWords = 16;       // targets of 5..8 symbols
constexpr int kMaxLongReadWords4 = 32;      // targets of up to 4 symbols
constexpr int kFilterFromWords = 13;        // HW queries of this many words and more take the piece filter (long_reads.hip)
inline int read_group_words(int m) { const int w = (m + 31) / 32; return w <= kMaxReadWords ? w : (w <= 10 ? 10 : (w <= 12 ? 12 : (w <= 14 ? 14 : (w <= 16 ? 16 : (w <= 24 ? 24 : 32))))); }
constexpr int kLanes = 64;            // wave64, hard-coded (gfx950)

// Everything the scan kernel needs; plain pointers into HBM.
struct ReadScanArgs {
    const uint32_t* peq;      // [readBlock][S symbols][NWD words][64 lanes], S = 4 (both kernels), 8 or 16 (banded kernel)
    const uint32_t* tpk;      // target, 2 bits / symbol, 16 symbols / dword, LSB first (scan_reads_kernel)
    const uint32_t* trows;    // target, 16 bits / symbol = LDS row offset (symbol << 8), 16 columns per 32-byte block,
                              // padded by two blocks (scan_reads_banded_kernel)
    int targetLength;
    const int* qlen;          // [slots] query length of the read in that slot (>= 1)
    const int* kinit;         // [slots] initial threshold: columns scoring <= kinit are candidates
    const int* slotmap;       // optional [lanes] lane -> slot indirection (second, exact pass)
    int nlanes;               // number of lanes to run (slots, or entries of slotmap)
    int numSegments;          // target split into this many segments (HW only; else 1)
    int segLen;               // columns per segment, multiple of 16
    int warm;                 // warm-up columns before a segment (2*maxQueryLen-1 for HW)
    int* segBest;             // [lanes][numSegments] best score seen in the segment (NW: final score)
    int* segCnt;              // [lanes][numSegments] number of columns attaining it
    int* segPos;              // positions pool
    int cap;                  // positions kept per (lane, segment) when posOff == nullptr
    const long long* posOff;  // optional [lanes][numSegments] offset into segPos (exact pass)
    const int* posCap;        // optional [lanes][numSegments] capacity (exact pass)
    int kcap;                 // banded HW kernel: effective threshold = min(kinit[slot], kcap)
    unsigned long long* wordSteps;   // banded HW kernel: += 32-row word-columns actually computed (may be null)
    // scan_reads_full_kernel, strips of a taller query (null: a whole query per lane).  One dword per 16 columns and lane, two
    // bits per column (bit 0: +1, bit 1: -1): the horizontal deltas of the bottom row of the strip above (in) / of this strip
    // (out), laid out [segment][block of 16 columns][lane of the producing launch]
    const uint32_t* chainIn; uint32_t* chainOut;
    const int* chainSrc;      // [lanes] lane of the producing launch that holds the strip above
    int chainInLanes;         // nlanes of the producing launch
    int chainBlocks;          // blocks of 16 columns per segment in the streams (>= (segLen + warm) / 16 + 2)
    const int* rowBase;       // [slots] query rows above the strip (its bottom row starts at score rowBase + qlen); null: 0
    int filter;               // banded HW kernel: 1 = fixed threshold, segPos lists the 16-column BLOCKS that hold a column
                              // scoring <= kinit (each once), segCnt their number (piece filter of long reads)
};

// mode: 0 NW, 1 SHW, 2 HW (values of EdlibAlignMode).  Returns hipSuccess or the launch error.
hipError_t launch_scan_reads(int nwords, int mode, const ReadScanArgs& a, hipStream_t stream);

// HW only: Ukkonen-banded variant with k-doubling (see reads_kernels.hip); same Peq layout.  syms = Peq rows per word:
// 4, 8 or 16 (target symbols rounded up); launch_scan_reads only knows 4.
hipError_t launch_scan_reads_banded(int nwords, int syms, const ReadScanArgs& a, hipStream_t stream);

// HW only, no band: every row of every column with the banded kernel's data path (syms = 4, 8 or 16)
hipError_t launch_scan_reads_full(int nwords, int syms, const ReadScanArgs& a, hipStream_t stream);

hipError_t launch_pack_target_2bit(const uint8_t* raw, const uint8_t* lut, int targetLength,
                                   uint32_t* tpk, hipStream_t stream);
// ndwords = 8 * (blocks of 16 columns, including the two blocks of padding): every dword is written
hipError_t launch_pack_target_rows(const uint8_t* raw, const uint8_t* lut, int targetLength,
                                   uint32_t* trows, int ndwords, hipStream_t stream);

// Builds Peq for every slot (reference buildPeq, edlib.cpp:358-384, for the <= syms target symbols; eqtbl[byte] =
// 16-bit set of the target symbols a query byte equals),
// the per-slot query length and the number of query byte values absent from the target.
hipError_t launch_build_peq_reads(int nwords, int syms, const uint8_t* reads, const long long* qoff,
                                  const int* perm, int nslots, const uint16_t* eqtbl,
                                  const uint32_t* targetPresence /*8 dwords*/, int kcfg,
                                  uint32_t* peq, int* qlen, int* kinit, int* alphaExtra,
                                  hipStream_t stream);

// Piece filter: gathers the (lane, block) candidates of a filter scan into one list.  out[2i] = lane, out[2i+1] = block;
// *counter = number of candidates (may exceed maxOut: the caller retries with a larger list); overflow[lane] = 1 when a
// (lane, segment) record held more than `cap` blocks.
hipError_t launch_collect_candidates(const int* segCnt, const int* segPos, int numSegments, int cap, int nlanes,
                                     int* out, int maxOut, int* counter, int* overflow, hipStream_t stream);

hipError_t launch_merge_segments(const int* segBest, const int* segCnt, const int* segPos,
                                 int numSegments, int cap, int nlanes, const int* slotmap, int capFinal,
                                 int* best, int* total, int* pos, int* flags, hipStream_t stream);

}  // namespace edlib_amd
