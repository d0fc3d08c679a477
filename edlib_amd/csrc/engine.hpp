// engine.hpp -- host side of the engine: batches of alignment units resident in
// HBM, scheduled onto the two kernel families.  This is the re-expression of the
// orchestration in reference edlibAlign() (edlib.cpp:146-301) for a batch.
#pragma once
#include "../../include/edlib_amd.h"
#include "common.hpp"
#include "pair_kernels.hpp"
#include "lanepair.hpp"
#include "reads_kernels.hpp"

#include <chrono>
#include <deque>
#include <functional>
#include <memory>
#include <vector>

namespace edlib_amd {

// Byte-level tables shared by every unit of a batch:
// the target alphabet and the equality relation of EqualityDefinition
// (edlib.cpp:63-94) expressed on raw bytes.
struct Tables {
    uint8_t tlut[256];        // target byte -> symbol id (first appearance order)
    uint8_t idToByte[256];    // symbol id -> byte
    int sigmaT = 0;           // number of distinct target bytes
    uint32_t presence[8];     // bitset of the target bytes
    uint16_t eqtbl[256];      // query byte -> 16-bit set of the (first 16) target symbols it equals
    std::vector<uint8_t> eq8; // [256*256] eq8[q*256+t] = 1 if bytes q and t are equal; EMPTY = identity
};

// A unit handed to the block-per-lane kernels (host mirror of PairDesc).
struct UnitSpec {
    long long qoff; int qlen; int qstep;
    long long toff; int tlen; int tstep;
    int kinit;
    int skip = 0;      // HW target segments: leading warm-up columns whose scores are not recorded
    int band = 0;      // SHW: scan only the diagonals [-kinit, kinit] (a cell within kinit of the origin's diagonal has |i - j| <= kinit);
                       // HW: only [-kinit, (tlen - qlen) + 2 kinit] (an alignment within kinit starts in [0, tlen - qlen + kinit])
};

struct SolveOut {
    std::vector<int> score, count, last;
    std::vector<long long> posStart;      // [units+1] into posFlat (exact, complete lists)
    std::vector<int> posFlat;
    // op strings stay where the D2H put them (pinned staging, one block per chunk); units point into it
    std::vector<const uint8_t*> opsPtr;
    std::vector<int> opsLen;
    std::vector<std::shared_ptr<PinBuf>> opsBufs;
};

// Op string of one job: a view into a staging block, or an owned concatenation (Hirschberg pieces).
struct OpsOut {
    const uint8_t* p = nullptr;
    int len = 0;
    std::vector<uint8_t> own;
};

// Location lists of a unit: almost always one or two entries, so they live inside the result record (a
// std::vector here was two malloc / free pairs per read: most of what LOC / PATH cost on a read batch); longer
// lists move to the heap.  24 bytes: the host side of LOC / PATH is bound by walking these records.
class LocList {
public:
    LocList() = default;
    ~LocList() { delete big_; }
    LocList(const LocList& o) { append(o.data(), o.n_); }
    LocList& operator=(const LocList& o) { if (this != &o) { clear(); append(o.data(), o.n_); } return *this; }
    LocList(LocList&& o) noexcept : n_(o.n_), big_(o.big_) { in_[0] = o.in_[0]; in_[1] = o.in_[1]; o.n_ = 0; o.big_ = nullptr; }
    LocList& operator=(LocList&& o) noexcept {
        if (this != &o) { delete big_; n_ = o.n_; big_ = o.big_; in_[0] = o.in_[0]; in_[1] = o.in_[1]; o.n_ = 0; o.big_ = nullptr; }
        return *this;
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    const int* data() const { return n_ <= kInline ? in_ : big_->data(); }
    int* data() { return n_ <= kInline ? in_ : big_->data(); }
    int operator[](size_t i) const { return data()[i]; }
    int& operator[](size_t i) { return data()[i]; }
    void clear() { n_ = 0; }                              // keeps the heap block (records are recycled across runs)
    void push_back(int v) { append(&v, 1); }
    void append(const int* p, size_t c) {
        const size_t nn = n_ + c;
        if (nn <= kInline) { for (size_t i = 0; i < c; ++i) in_[n_ + i] = p[i]; }
        else {
            if (!big_) big_ = new std::vector<int>();
            if (n_ <= kInline) big_->assign(in_, in_ + n_);
            big_->insert(big_->end(), p, p + c);
        }
        n_ = (uint32_t)nn;
    }
    void assign(size_t c, int v) {
        n_ = 0;
        if (c <= kInline) { for (size_t i = 0; i < c; ++i) in_[i] = v; }
        else { if (!big_) big_ = new std::vector<int>(); big_->assign(c, v); }
        n_ = (uint32_t)c;
    }
private:
    static const uint32_t kInline = 2;
    uint32_t n_ = 0;
    int in_[kInline];
    std::vector<int>* big_ = nullptr;
};

// Per-unit result assembled on the host before it is marshalled into
// EdlibAlignResult (malloc'd arrays) by results().
struct UnitResult {
    int status = EDLIB_STATUS_OK;
    int editDistance = -1;
    int alphabetLength = 0;
    bool hasEnds = false, hasStarts = false, hasAlignment = false;
    LocList ends, starts;
    const uint8_t* opsView = nullptr;         // op string: a view into a staging block (or an owned string) the batch keeps alive
    int opsViewLen = 0;
};

class Batch {
public:
    virtual ~Batch();
    // sequences are copied to the device here; nothing of the caller's memory is retained
    int init(const char* queries, const long long* qoff, int n,
             const char* targets, const long long* toff, int numTargets,   // numTargets == 1: shared
             EdlibAlignConfig cfg, int device);
    int run();
    int runImpl();
    int results(EdlibAlignResult* out);
    int resultsFlat(int* status, int* editDistance, int* numLocations, int* alphabetLength, long long* locOffsets,
                    int** endLocations, int** startLocations, long long* alnOffsets, unsigned char** alignment);
    int resultsView(EdlibAmdResultsView* out);
    int cigarView(int format, const char** chars, const long long** offsets);
    EdlibAmdBatchStats stats{};
    void finishStats();          // fills the fields of `stats` that cost a walk over the records (algo_bytes)

private:
    // ---- configuration
    EdlibAlignConfig cfg_{};
    std::vector<EdlibEqualityPair> eqs_;
    int device_ = 0;
    bool shared_ = false;
    int n_ = 0;
    std::vector<long long> qoff_, toff_;
    Tables tab_;
    hipStream_t stream_ = nullptr;
    Event evRun0_, evRun1_, evA_, evB_;           // evB_: the side stream's alphabet kernel is done

    // ---- resident inputs
    DevBuf<uint8_t> d_in_;                       // offsets + small tables (+ small sequence pools) in one block
    PinBuf h_in_;                                // its pinned staging (kept until the batch dies: the copy is asynchronous)
    DevBuf<uint8_t> d_qpool_, d_tpool_, d_tlut_, d_idToByte_, d_eq8_;
    DevBuf<uint16_t> d_eqtbl_;   // views into d_in_ unless large
    hipError_t uploadEq8();
    DevBuf<uint32_t> d_presence_;
    DevBuf<long long> d_qoff_, d_toff_;

    // ---- unit classification (phase 1)
    std::vector<int> emptyUnits_, readUnits_, pairUnits_;
    std::vector<int> longUnits_;                 // HW queries above 256 rows against the shared target (long_reads.hip)
    std::vector<int> pairNow_;                   // pair units of the current run: pairUnits_ + what the piece filter handed back

    // ---- reads-per-lane path
    struct ReadGroup {
        int nwords = 0, nslots = 0, numSegments = 1, segLen = 0, warm = 0;
        std::vector<int> perm;                 // slot -> unit (or -1)
        DevBuf<int> d_perm, d_qlen, d_kinit, d_alphaExtra, d_segBest, d_segCnt, d_segPos;
        DevBuf<int> d_best, d_total, d_pos, d_flags;
        DevBuf<uint32_t> d_peq;
        bool zeroCopy = false; PinBuf hostOut;   // small groups: d_best / d_total / d_alphaExtra / d_flags / d_pos are views into pinned host memory
        // exact second pass of the last run: overflowing slots, their offsets into d_ovfPool
        std::vector<int> ovfSlots; std::vector<long long> ovfOff; DevBuf<int> d_ovfPool;
    };
    std::vector<std::unique_ptr<ReadGroup>> groups_;
    DevBuf<uint32_t> d_tpk_, d_trows_;
    DevBuf<unsigned long long> d_wordSteps_;
    PinBuf h_wordSteps_; bool wordStepsPending_ = false;
    DevBuf<unsigned long long> d_ringSteps_; PinBuf h_ringSteps_; bool ringStepsUsed_ = false;   // live word-steps of the ring kernels of a run
    unsigned long long* ringStepsCounter();
    bool banded_ = false;        // HW groups use the Ukkonen-banded kernel with k-doubling
    int syms_ = 4;               // Peq rows per word of the reads kernels: target symbols rounded up to 4, 8 or 16
    int packTarget();
    int runReads();                                   // device work only; results stay in HBM
    int makeGroup(const std::vector<int>& units, int words, std::unique_ptr<ReadGroup>& g, long long oneRoundWaves = 0);
    int runGroupScans(ReadGroup& g, bool fullOnly);
    int runGroupExact(ReadGroup& g);
    int collectGroup(ReadGroup& g, std::vector<UnitResult>& res);
    // ---- long HW queries: piece filter on the reads-per-lane kernel + window verification on kernel W (long_reads.hip)
    struct Piece { long long off; int len; int thr; };           // rows [off, off + len) of the query pool, threshold of its scan
    int scanPieces(const std::vector<Piece>& pieces, bool filter, std::vector<std::pair<int, int>>* cand,
                   std::vector<uint8_t>* overflow, std::vector<int>* best);
    int solveLongReads(std::vector<UnitResult>& res, std::vector<int>& fallback);
    bool filterScan_ = false;
    // strips of taller queries on the full-height lane kernel, chained through HBM (long_reads.hip: solveTallFull)
    struct ChainArgs { const uint32_t* in = nullptr; uint32_t* out = nullptr; const int* src = nullptr; int inLanes = 0, blocks = 0; const int* rowBase = nullptr; };
    ChainArgs chain_;
    int solveTallFull(const std::vector<int>& units, std::vector<UnitResult>& res, std::vector<int>& handBack);
    int collectReads(std::vector<UnitResult>& res);   // D2H + result semantics (lazy for TASK_DISTANCE)
    bool readsCollected_ = true;
    int scanGroup(ReadGroup& g, int mode, const int* d_slotmap, int nlanes, int kcap, const int* d_kinit,
                  int numSegments, int segLen, int warm, int* segBest, int* segCnt, int* segPos, int cap,
                  const long long* posOff, const int* posCap, bool unbanded = false,
                  unsigned long long* wordSteps = nullptr, const uint32_t* peqDense = nullptr,
                  const int* qlenDense = nullptr);

    // ---- block-per-lane path
    DevBuf<PairDesc> d_descs_;
    DevBuf<unsigned long long> d_peq64_;
    DevBuf<StoreEntry> d_store_;
    DevBuf<int> d_aux_, d_out3_, d_outScore_, d_outCount_, d_outLast_, d_posPool_, d_opsLen_, d_alpha_;   // d_out{Score,Count,Last}_: views into d_out3_
    DevBuf<uint8_t> d_ops_;
    DevBuf<long long> d_opsOff_;
    // ring = 0: unbanded strips (any mode).  ring = 4 / 16 / 64: NW inside Ukkonen's band for threshold
    // UnitSpec::kinit on rings of that many lanes (exact iff score <= kinit); the caller guarantees that
    // the band fits (kinit <= ring_max_k(ring), or the unit has at most `ring` blocks)
    // ringH = 2 / 4 (ring 16, no path): ring lanes of 2 / 4 vertically adjacent blocks, band limit ring_max_k(16, ringH)
    int solve(int mode, bool wantPositions, bool wantPath, const std::vector<UnitSpec>& units, SolveOut& out,
              int ring = 0, int ringH = 1);
    int solveChunk(int mode, bool wantPositions, bool wantPath, const std::vector<UnitSpec>& units,
                   size_t a, size_t b, SolveOut& out, int ring, int ringH);
    // ---- one unit on many waves (wide_kernels.hip): `ring` = kWide in solve() / solveChunk().  NW: the band of threshold
    // UnitSpec::kinit, any width (exact iff score <= kinit); SHW / HW: the pipelined strips.  Distances / positions only.
    static const int kWide = -1;
    static const int kRing32 = -2;                // solve(): storing NW scans on 16-lane rings of 32-row words (the caller checked ring32_fits)
    bool ring32_fits(const std::vector<UnitSpec>& units, size_t a, size_t b, int G) const;
    DevBuf<unsigned long long> d_wide_; DevBuf<unsigned> d_wabort_; PinBuf h_wabort_;
    int wideCap_ = -1;                           // resident waves of the wide kernel on this device
    struct WidePlan { int slots = 1; size_t perLaunch = 1; };
    int planWide(int mode, PairDesc* descs, size_t n, WidePlan& plan);      // slots per unit, units per launch; assigns auxOff
    int launchWide(int mode, const PairScanArgs& a, const PairDesc* hostDescs, size_t n, const WidePlan& plan);
    int checkWide();                             // after the stream is idle: 0 fine, 2 a launch gave up (run the units again: one slot each), 1 error.  Releases the device's wide gate.
    bool wideSerial_ = false;                    // this run's wide launches take one slot per unit (after an aborted launch)
    bool wideGateHeld_ = false;                  // this batch holds its device's wide gate (launchWide .. checkWide)
    void wideGateRelease();
    // NW distance of long units as two half scans that meet in the middle (forward over the left half of the target,
    // reverse over the right half, both inside the band of UnitSpec::kinit): out[4 u ..] = {min, split row, left, right};
    // exact iff min <= kinit
    int solveWideSplit(const std::vector<UnitSpec>& units, std::vector<int>& out, int ring = 0);     // ring > 0: the halves on lane rings
    // what the distance phase of this run already knows about a piece's first split (solveWideSplit): the path phase
    // does not scan the two halves again
    struct KnownSplit { long long qoff; int m; long long toff; int T; int score; int row, left, right; };
    std::vector<KnownSplit> knownSplits_;
    // NW distances by threshold levels on rings of 4, 16, 64 lanes, then unbanded (the reference's
    // k-doubling, edlib.cpp:197-217, with thresholds chosen for the hardware)
    // paths != null (TASK_PATH, every unit below the 1 MiB rule): the levels run with the column store and the traceback, so
    // that a unit's first successful level is also its path (one scan instead of the distance scan + the storing scan)
    bool hwBandSplit_ = false;                   // solveSemiGlobal: the units its HW band does not take are on their way through it again
    int solveGlobalDistances(const std::vector<UnitSpec>& units, std::vector<int>& score, std::vector<OpsOut>* paths = nullptr);
    // SHW / HW units: short queries packed on 4- and 16-lane rings, the rest on the strips
    int solveSemiGlobal(int mode, bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out);
    int solveSemiGlobalUnits(int mode, bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out);
    // SHW inside Ukkonen's band (edlib.cpp:562, 602-630): threshold levels like the NW distances, target cut at m + K
    int solveShwBanded(bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out);
    // HW inside the static band of a threshold: diagonals [-K, (T - m) + 2 K] (queries in windows not much longer than themselves)
    int solveHwBanded(bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out);
    // alphabetLength of the empty / pair units: launched on a side stream before phase 1, collected after it
    int alphabetLengthsBegin(hipEvent_t after = nullptr);      // after: the side stream also waits for this event
    bool alphaDeferred_ = false;
    int alphabetLengthsEnd(std::vector<UnitResult>& res);
    std::vector<int> alphaUnits_; bool alphaOnHost_ = false, alphaPending_ = false; long long alphaBytes_ = 0;
    hipStream_t side_ = nullptr;
    DevBuf<int> d_alphaIdx_, d_alphaOut_; PinBuf alphaPin_;
    // linear-space paths (reference obtainAlignmentHirschberg, edlib.cpp:1231-1396)
    struct PathPiece { long long qoff; int m; long long toff; int T; int score; };
    int hirschbergLevel(const std::vector<PathPiece>& big, std::vector<int>& splitRow,
                        std::vector<int>& leftScore, std::vector<int>& rightScore);
    int solvePaths(const std::vector<PathPiece>& jobs, std::vector<OpsOut>& opsOut, std::vector<int>& status);
    std::vector<std::shared_ptr<PinBuf>> opsKeep_;      // staging blocks the last run's op views point into
    std::deque<std::vector<uint8_t>> opsOwned_;         // op strings assembled on the host (Hirschberg pieces, empty windows)

    int qlen(int u) const { return (int)(qoff_[u + 1] - qoff_[u]); }
    long long tbase(int u) const { return shared_ ? toff_[0] : toff_[u]; }
    int tlen(int u) const { return shared_ ? (int)(toff_[1] - toff_[0]) : (int)(toff_[u + 1] - toff_[u]); }

    void scanTimerStart();
    void scanTimerStop();
    std::vector<std::pair<hipEvent_t, hipEvent_t>> scanEvents_;
    size_t scanEventsUsed_ = 0;

    std::vector<UnitResult> results_, work_;     // last run's records / the next run's (ping-pong: no reallocation per run)
    std::vector<UnitSpec> startUnits_; std::vector<std::pair<int, int>> startWhere_; std::vector<int> live_;
    bool haveResults_ = false;
    long long algoBase_ = -1;     // algorithmic bytes of the batch while its results are still on the device
    bool algoDirty_ = false;
    // per-unit scratch of the pair path, kept across runs (fresh multi-megabyte vectors are mmap + page faults + munmap)
    std::vector<int> pairSpecsFor_;              // the pair units pairSpecs_ was built for
    std::vector<UnitSpec> pairSpecs_, selScratch_; std::vector<size_t> whoScratch_; std::vector<int> lvlScratch_, scoreMain_;
    SolveOut soMain_, soLevel_;
    // a ladder level that takes every unit of a big pair batch (engine_pairs.hip): compact specs resident per batch,
    // descriptors written by a kernel, the Peq of all units built on a third stream while the divergence probe runs
    int pairSpecsVersion_ = 0, levelSpecsVersion_ = -1, shapeVersion_ = -1, alphaIsPairsVersion_ = -1;
    bool alphaIsPairs_ = false;
    struct PairShape { size_t minBlocks = 0, maxBlocks = 0; int minLenLo = 0, minLenHi = 0, maxDiff = 0; } shape_;   // of pairSpecs_
    long long levelPeqWords_ = 0;
    bool levelAllReady_ = false;
    PinBuf h_levelSpecs_, h_levelScore_;
    DevBuf<LevelSpec> d_levelSpecs_;
    DevBuf<PairDesc> d_descsAll_;
    DevBuf<unsigned long long> d_peqAll_;
    hipStream_t aux_ = nullptr;
    Event evLevelIn_, evLevelPeq_;
    int prepareLevelAll(const std::vector<UnitSpec>& units);
    std::function<void()> whileScanning_;        // host work of the run that does not depend on the scan: done behind the level's launch
    int runLevelAll(const std::vector<UnitSpec>& units, int ring, int ringH, int ringBlocks, int cap, int kcap, int nbMax);
    // ---- the lane-per-pair level of big NW distance batches over at most four target symbols (lanepair.hpp):
    // a lane owns a unit, the band narrows with the scores.  The units' packed layout is resident like the offset arrays;
    // a run packs the sequences (query / target bit planes; alphabetLength counted on the way) beside the divergence probe
    // and scans every unit at the probe's threshold; what that leaves open climbs the rings.
    int laneSpecsVersion_ = -1;
    long long lanePlaneWords_ = -1, laneTgtWords_ = 0;
    bool laneReady_ = false;
    PinBuf h_laneUnits_, h_laneScore_;
    DevBuf<lanepair::LaneUnit> d_laneUnits_;
    DevBuf<lanepair::Plane2> d_lanePlanes_;
    DevBuf<lanepair::Tgt2> d_laneTgts_;
    DevBuf<int> d_laneFlags_, d_laneScore_;
    Event evLanePack_;
    int alphabetBuffers();
    int prepareLaneLevel(const std::vector<UnitSpec>& units);
    int runLaneLevel(const std::vector<UnitSpec>& units, double rate, int kcap, int W);
    std::vector<OpsOut> fusedOps_;
    // ---- flat pair path (TASK_DISTANCE, every unit a pair of at most 16 blocks): descriptors built once and resident, a
    // run is Peq build + ONE ring scan + an overflow census, results stay in HBM until results() (like the reads path)
    bool flatPairs_ = false, pairsCollected_ = true, lastRunFlat_ = false;
    int flatRing_ = 0;
    long long flatWordSteps_ = 0;               // word-steps inside the bands of one run over the flat descriptors
    DevBuf<PairDesc> d_flatDescs_;
    DevBuf<int> d_flatOut3_, d_flatPos_, d_flatCensus_;
    PinBuf h_flatCensus_;
    std::vector<long long> flatPeqOff_;
    std::vector<int> flatOvfUnit_; std::vector<long long> flatOvfOff_; std::vector<int> flatOvfPos_;   // exact lists of the last run's overflowing units
    PairDesc flatDesc(int u) const;
    int initFlatPairs();
    int runPairsFlat(bool& overflowed, bool& fellBack);
    // flat LOC / PATH (round 4): start locations and paths of a flat batch are found and kept on the device as well.
    // HW starts: one reverse prefix scan per end location, descriptors written by a kernel from the phase-1 results;
    // paths: one storing scan per unit over its window + the traceback into resident op slots.  What the fixed layouts
    // cannot hold (more than 16 end locations, a band level that fails) makes the run fall back to the general path.
    bool flatStarts_ = false, flatPaths_ = false, flatNwStore_ = false;
    int flatMaxBlocks_ = 0;
    size_t flatStartCap_ = 0;
    long long flatRevPeqBase_ = 0, flatOpsTotal_ = 0;
    DevBuf<PairDesc> d_flatRevDescs_, d_flatStartDescs_, d_flatPathDescs_;
    DevBuf<int> d_flatSlotOf_, d_flatStartOut3_, d_flatStartsOut_, d_flatPathOut3_, d_flatOpsLen_;
    DevBuf<long long> d_flatOpsOff_, d_flatStoreBase_;
    DevBuf<uint8_t> d_flatOps_;
    std::vector<long long> flatOpsOffHost_;
    int runFlatStartsAndPaths(bool& fellBack);
    int collectPairsFlat(std::vector<UnitResult>& res);
    bool readsViewOnDevice() const;               // a DISTANCE run over read groups only: its view is made by flat_results.hip
    int buildReadsView();
    // rings of 32-row words for the storing scans and walks of a flat PATH batch (ring32_kernels.hip)
    bool flatRing32_ = false; int flatG32_ = 8, flatMaxWords_ = 0;
    bool deferReadsReset_ = false;                // this run blanks the recycled records of reads-path units where it fills them (collectGroup)
    std::vector<int> flatChunkStart_;             // ring32 NW store: unit ranges whose store fits 32-bit offsets (one range = the usual case)
    DevBuf<uint8_t> d_tsym_;
    // the caller-facing arrays of the last run: made on the device for a flat batch (buildFlatView), from the records otherwise
    DevBuf<uint8_t> d_view_; PinBuf h_view_, h_viewVar_; bool viewReady_ = false;     // host: per-unit fields + offsets / locations + op bytes
    EdlibAmdResultsView view_{};
    const uint8_t* viewAlnDev_ = nullptr; const long long* viewAlnOffDev_ = nullptr;     // the dense op bytes on the device (CIGARs)
    std::vector<int> viewInts_; std::vector<long long> viewOffs_; std::vector<uint8_t> viewOps_;   // host-made view of a general batch
    DevBuf<int> d_flatOvfAt_, d_flatOvfPool_; DevBuf<long long> d_flatOvfOff_;
    int buildFlatView();
    int buildHostView();
    // CIGAR strings of the last run (edlibAlignmentToCigar, edlib.cpp:303-350, over the batch): [0] extended, [1] standard
    struct CigarOut { bool ready = false; PinBuf chars, offs; std::vector<char> hostChars; std::vector<long long> hostOffs; const char* p = nullptr; const long long* off = nullptr; };
    CigarOut cigar_[2];
    DevBuf<long long> d_cigWork_; DevBuf<char> d_cigChars_, d_cigChars2_;
    bool cigarSticky_ = false;                   // a caller of this batch has asked for CIGARs: later collections make them while the op bytes travel
    Event evView_;
    int enqueueCigars(int f, const uint8_t* aln, const long long* alnOff, size_t cap, hipStream_t st);
    int fetchCigars(int f, hipStream_t st);
    int ensureCollected();                       // results of the last run that are still on the device -> results_
};

// single-pair convenience used by edlibAlign()
int align_one(const char* q, int qn, const char* t, int tn, EdlibAlignConfig cfg, EdlibAlignResult* out);
// one small pair in one kernel launch (one_pair.hip): 0 = answered, 1 = error, 2 = not handled here (take align_one)
int align_one_fused(const char* q, int qn, const char* t, int tn, EdlibAlignConfig cfg, EdlibAlignResult* out);

// EDLIB_AMD_DEBUG: host wall time between named points of a run (stderr)
struct Lap {
    bool on; std::chrono::steady_clock::time_point t;
    Lap() : on(getenv("EDLIB_AMD_DEBUG") != nullptr), t(std::chrono::steady_clock::now()) {}
    void operator()(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[edlib_amd] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// helpers shared by the translation units of the host side
const int kMaxDevices = 16;                       // devices the process-wide caches and gates are kept for
const int kPosCap = 16;                           // end positions kept beside a pair unit's results (longer lists: exact second pass)
int peq_row_stride(long long nb);                 // row length of the LDS-resident Peq of the ring kernels
bool needs_hirschberg(int m, int T);              // the reference's 1 MiB rule (edlib.cpp:1188-1190)
void blank_record(UnitResult& r);
void finalize_global(UnitResult& r, int kcfg, int mode, int T, int score);
long long hw_band_rows(const UnitSpec& u, long long K);     // rows per column of the HW band of threshold K
// helpers shared by engine.hip and long_reads.hip
int roundup(int x, int q);
void plan_segments(int nlanes, int T, int mode, int warmFull, long long wantWaves, int& S, int& segLen, int& warm, int minSegCols = 4096);
// waves of the 24- / 32-word full-height lane kernels (24 / 32 KB of LDS rows each) the chip runs without two of them sharing
// a SIMD: one per SIMD of 256 CUs (swept in round 4: DESIGN.md 3c)
inline long long tall_round_waves() { return 1024; }
void finalize_semiglobal(UnitResult& r, int kcfg, int m, int best, const int* pos, long long npos);

int device_count();
int host_threads(int cap);
int default_device();

}  // namespace edlib_amd
