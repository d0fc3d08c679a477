// pair_kernels.hpp -- launch interface of the "block-per-lane" family: one
// wave64 owns one (query, target) unit, lanes are 64-row blocks of the query
// column, the horizontal carry travels lane -> lane+1 by DPP.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace edlib_amd {

// One unit of work.  Sequences are addressed inside device-resident byte pools;
// step = -1 walks a sequence backwards (used for the reverse scans that find
// HW start locations, reference edlib.cpp:230-266, without materialising
// reversed copies).
struct PairDesc {
    long long qoff;      // first byte of the query in the query pool (last byte if qstep < 0)
    long long toff;      // same for the target
    long long peqOff;    // first word of this unit's Peq in the Peq pool: [sym][numBlocks]
    long long storeOff;  // first entry of this unit's column store (PATH), in block-steps
    long long auxOff;    // first int of this unit's strip hand-off buffer (targetLen ints)
    int qlen;
    int tlen;
    int qstep;           // +1 / -1
    int tstep;           // +1 / -1
    int kinit;           // SHW/HW: columns scoring <= kinit are end-location candidates
    int posCap;          // capacity of this unit's end-position list
    long long posOff;    // first int of that list in the positions pool
    long long colOff;    // first block of this unit's last-column dump (Hirschberg), or -1
    int bandT;           // banded NW kernel: target length that defines the band when the scan stops early
                         // at column tlen-1 (Hirschberg halves, edlib.cpp:1252-1260); 0 = tlen
    int skip;            // SHW / HW tracking: columns before this one are warm-up and record nothing (HW target segments)
    int ring;            // layout of this unit's column store: 0 = strips (scan_pairs_kernel), else the ring
                         // size G of scan_pairs_ring_kernel (band of threshold kinit)
};

// A pair batch's units as the device keeps them between runs: offsets and lengths of the resident inputs and the unit's slot
// in a Peq pool that holds every unit.  A level of the NW threshold ladder that takes EVERY unit of a big batch (config 4:
// 100,000 pairs on 21-lane rings) has its descriptors written from these by a kernel -- no host loop over the units, no
// 9 MB upload -- and finds its Peq built while the divergence probe ran (Batch::prepareLevelAll / runLevelAll).
struct LevelSpec { long long qoff, toff, peqOff; int qlen, tlen; };
// descriptors of forward NW distance units on `ring`-lane rings: kinit = min(kcap, the whole matrix when the unit's blocks
// all sit on the ring (numBlocks <= ringBlocks), else cap)
hipError_t launch_fill_level_descs(const LevelSpec* specs, int numUnits, int kcap, int ringBlocks, int cap, int ring,
                                   PairDesc* out, hipStream_t stream);

// One block-step of the column store: what the traceback needs to know about the 64 cells of block b in column c, as two
// bit planes (bit r = row 64 b + r).  The reference keeps Pv, Mv and the block score per column (AlignmentData,
// edlib.cpp:22-47: 20 bytes) and re-derives the neighbours' values cell by cell (edlib.cpp:942-1141); the walk only ever
// asks three questions of a cell, in this order (up > left > diagonal):
//   up possible    <=> D[r][c] = D[r-1][c] + 1    <=> Pv bit (vertical delta +1, after the column)
//   left possible  <=> D[r][c] = D[r][c-1] + 1    <=> Ph bit (horizontal delta +1, before the shift; calculateBlock :426)
//   diagonal free  <=> D[r][c] = D[r-1][c-1]      <=> Xh bit when neither of the above holds (:424; Xh | Mv is Hyyro's D0,
//                                                     and a set Mv bit of column c-1 makes Ph = 1, i.e. the walk goes left)
// Four outcomes per cell = two bits:  x = Pv | Ph ("an indel move"),  y = ~Pv & (Ph | Xh):
//   up = x & ~y,  left = x & y,  diagonal = ~x with MATCH iff y.   16 bytes: one store per block-step, one load per column.
struct __attribute__((aligned(16))) StoreEntry {
    unsigned long long x, y;
};

struct PairScanArgs {
    const PairDesc* descs;
    int numUnits;
    const uint8_t* qpool;
    const uint8_t* tpool;
    const uint8_t* tlut;        // [256] target byte -> symbol id (row of Peq)
    const uint8_t* tsym;        // ring32 kernels: the target pool as symbol ids (launch_target_symbols), same offsets as tpool
    int sigmaT;                 // number of target symbols (rows of Peq)
    const unsigned long long* peq;   // Peq pool, built by launch_build_peq_pairs
    int peqFullStride;          // ring kernel: sigmaT * peqRowStride, 0 = unknown
    int peqRowStride;           // largest block count of the launch, padded (see scan_pairs_ring_kernel)
    int* aux;                   // strip hand-off pool (horizontal deltas of a strip's bottom row)
    // column store for the traceback (may be null), see pair_kernels.hip for the two layouts
    StoreEntry* store;
    // outputs
    int* outScore;              // [units] NW: D[m][T]; SHW/HW: best bottom-row score (or -1)
    int* outCount;              // [units] SHW/HW: number of columns attaining it
    int* outLast;               // [units] SHW/HW: last (largest) such column, -1 if none
    int* posPool;               // end positions
    // last-column dump (may be null): (Pv, Mv, block score) of every block at column tlen-1
    unsigned long long* colP;
    unsigned long long* colM;
    int* colS;
    // ring kernel: += 32-row word-columns of blocks INSIDE their life (the band), i.e. without the updates a lane runs on
    // dead state between two blocks (may be null)
    unsigned long long* wordSteps;
    // wide kernel (wide_kernels.hip): hand-off granules of the strip pipelines (PairDesc::auxOff = the unit's first granule,
    // wide_stream_words() per unit, zeroed before every launch) and the launch's abort word
    unsigned long long* wstream;
    unsigned* wabort;           // {abort word, workgroups arrived}
    unsigned wideExpect;        // workgroups the residency check waits for; 0 = the grid (a test passes one more: the check must fail)
};

// mode: 0 NW, 1 SHW, 2 HW.  store: also write the column store.
hipError_t launch_scan_pairs(int mode, bool store, const PairScanArgs& a, hipStream_t stream);

// NW with Ukkonen's diagonal band for threshold desc.kinit (reference myersCalcEditDistanceNW with a
// fixed k, edlib.cpp:730-928): exact whenever the distance is <= kinit, otherwise some value > kinit.
// ringLanes G in {4, 8, 16, 21, 32, 64}: the band must fit the ring (kinit <= ring_max_k(G), or numBlocks <= G and any
// kinit); a wave carries 64 / G units, whatever the query length (no strips).  Writes outScore and, when
// colP is set, the (P, M, score) of the blocks alive at the last processed column (the caller pre-fills
// the dump with "invalid").  store: also the column store in ring layout (ring_store_entries per unit).
// Block b (64 H rows) is updated at steps [(64 H + 1) b + dmin, (64 H + 1) b + 64 H - 1 + dmax]: the next tenant of its lane,
// block b + G, starts (64 H + 1) G steps later, so a band of dmax - dmin <= (64 H + 1) G - 64 H diagonals keeps the tenants
// of a lane apart (tests/ring_model.py: ring_fits, ring_lanes_nw).  A whole-wave ring (G = 64) keeps one lane idle: with
// 64 blocks alive the 256-column target ring of the kernel would be refilled over the column its oldest block starts on.
constexpr int ring_max_k(int G, int H = 1) { return (64 * H + 1) * (G == 64 ? G - 1 : G) - 64 * H; }   // H: 64-row blocks per ring lane (1, 2 or 4)
constexpr int kNumRings = 6;           // ring sizes 4, 8, 16, 21, 32, 64 (units per wave: 16, 8, 4, 3, 2, 1)
constexpr int kMaxBandK = ring_max_k(64);
// mode 1 (SHW) / 2 (HW): packed rings only (ringLanes 4, 8 or 16); every unit has numBlocks <= ringLanes (no band, kinit is
// the end-location threshold) or desc.bandT < 0: the static band of threshold kinit -- SHW the diagonals [-K, K], HW
// [-K, (tlen - qlen) + 2 K] -- which must fit the ring like an NW band of that many rows; outputs as launch_scan_pairs.
// blocksPerLane H = 2 / 4 (16-lane rings, no store): a ring lane holds H vertically adjacent blocks ("superblock"); the band
// limit is ring_max_k(16, H), modes 1 / 2 take units of up to 16 H blocks.
hipError_t launch_scan_pairs_ring(int ringLanes, int mode, bool store, const PairScanArgs& a, hipStream_t stream,
                                  int blocksPerLane = 1);
long long ring_store_entries(int ringLanes, int qlen, int tlen);
// adds the word-steps inside the bands of `numUnits` ring units (64 H rows per ring-lane block, scan mode `mode`) to *out:
// what launch_scan_pairs_ring does behind its scan when PairScanArgs::wordSteps is set
hipError_t launch_count_ring_steps(const PairDesc* descs, int numUnits, int mode, int H, unsigned long long* out, hipStream_t stream);

// One unit on many waves (wide_kernels.hip): strips of 64 blocks as a pipeline over `slots` resident single-wave workgroups
// per unit (grid = slots x units: the caller keeps slots * units <= wide_resident_waves()).  mode 0: NW inside the band
// of threshold desc.kinit (any K >= |tlen - qlen|, any query length; exact iff the result is <= K), desc.bandT / colOff
// as for the rings; modes 1 / 2: every block of every column, outputs as launch_scan_pairs.  No column store.
hipError_t launch_scan_pairs_wide(int mode, const PairScanArgs& a, int slots, hipStream_t stream);
long long wide_stream_words(int tlen, int slots);                     // granules (u64) of one unit's hand-off area
int wide_slots_wanted(int mode, int qlen, int tlen, int bandT, int K);   // waves that keep the unit's pipeline from stalling
long long wide_word_steps(int mode, int qlen, int tlen, int bandT, int K);   // 32-row word-columns the launch computes
int wide_resident_waves(int sigmaT);

// reference buildPeq (edlib.cpp:358-384) for every unit: Peq[sym][block] from the
// query bytes and the 256x256 byte equality matrix eq8 (identity + additionalEqualities).
hipError_t launch_build_peq_pairs(const PairDesc* descs, int numUnits, const uint8_t* qpool,
                                  const uint8_t* eq8, const uint8_t* idToByte, int sigmaT,
                                  unsigned long long* peq, hipStream_t stream);

// ---- lane rings of 32-row words (ring32_kernels.hip): NW pairs of a batch too small to fill the chip (lone waves: bound
// by the latency of their own instruction stream), with the column store as 8-byte entries (the two planes of StoreEntry on
// 32 rows) in lines of eight steps per ring lane ([group of 8 steps][ring lane][step & 7] behind storeOff, in the u64 view
// of PairScanArgs::store; step = column + word), and a lane-parallel walk.
// G in {4, 8, 16}; the band of desc.kinit must fit the ring (kinit <= ring32_max_k(G)) or all words sit on it (any kinit).
// Forward units only (qstep = tstep = 1, no bandT / colOff); needs PairScanArgs::tsym; a launch's store stays below 4 GB.
constexpr int ring32_max_k(int G) { return 32 * (G - 2); }
long long ring32_store_entries(int ringLanes, int qlen, int tlen);          // u64 entries of one unit
size_t ring32_lds_bytes(int ringLanes, int sigmaT, int maxWords);           // dynamic LDS of a wave (the launcher refuses > 48 KB)
long long ring32_word_steps(int ringLanes, const PairDesc* hostDescs, int n);   // 32-row word-columns inside the bands (host)
hipError_t launch_target_symbols(const uint8_t* tpool, const uint8_t* tlut, long long n, uint8_t* tsym, hipStream_t stream);
hipError_t launch_scan_pairs_ring32(int ringLanes, bool store, const PairScanArgs& a, int maxWords, hipStream_t stream);

struct TracebackArgs {
    const PairDesc* descs;
    int numUnits;
    const int* score;           // [units] D[m][T] (start value of the walk)
    const StoreEntry* store;
    uint8_t* ops;               // ops pool; unit u owns [opsOff[u], opsOff[u + 1]), filled from the back
    const long long* opsOff;
    int* opsLen;                // [units] number of ops; they occupy the END of the unit's range
};
// reference obtainAlignmentTraceback (edlib.cpp:942-1141)
hipError_t launch_traceback(const TracebackArgs& a, hipStream_t stream);
// the same walk on the store of launch_scan_pairs_ring32 (store entries of 8 bytes, PairDesc::storeOff in those)
hipError_t launch_traceback32(const TracebackArgs& a, int ringLanes, hipStream_t stream);

// number of block-steps the column store of a unit needs
long long pair_store_entries(int qlen, int tlen);

// Hirschberg split (reference obtainAlignmentHirschberg, edlib.cpp:1314-1353).  For piece p the
// forward unit 2p holds the last column of (query vs left half), the reverse unit 2p+1 the last column
// of (reversed query vs reversed right half).  Finds the first query row i in [0, m-2] with
// L[i] + R[i+1] == best, else the boundary cases i = -1 / i = m-1; out[3p..3p+2] = {i, leftScore,
// rightScore} (i = -2 if nothing adds up: internal error).
struct SplitArgs {
    const PairDesc* descs;      // 2 per piece (forward, reverse)
    int numPieces;
    const int* best;            // [pieces] distance of the piece
    const unsigned long long* colP;
    const unsigned long long* colM;
    const int* colS;
    int* out;                   // [pieces][3]
};
hipError_t launch_hirschberg_split(const SplitArgs& a, hipStream_t stream);
// distance of a piece from the same two dumps: min over the rows of L[i] + R[i+1] (SplitArgs::best unused);
// out[4p..4p+3] = {min, row as above, leftScore, rightScore}; `packed`: numPieces words of scratch
hipError_t launch_split_min(const SplitArgs& a, unsigned long long* packed, int maxRows, hipStream_t stream);

}  // namespace edlib_amd
