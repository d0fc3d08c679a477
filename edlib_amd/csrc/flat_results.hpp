// flat_results.hpp -- launch interface of flat_results.hip: the results of a flat pair batch as dense caller-facing
// arrays, made on the device; CIGAR strings of a whole batch.
#pragma once
#include "pair_kernels.hpp"

namespace edlib_amd {

struct FlatResultArgs {
    const PairDesc* descs;      // the batch's resident descriptors (qlen, tlen); null: a batch of reads against one shared target --
    const int* qlens; int sharedT;   // -- whose query lengths are an array and whose target length is one number
    int alphaBase;              // added to alphabet[u] (the reads path counts what a query adds to the target's alphabet)
    int n, mode, k, wantPath, posCap;
    // what the scans left: score / count per unit, posCap end positions per unit, HW start locations (or null)
    const int* score; const int* count; const int* pos; const int* devStarts;
    // units of the exact second pass: ovfAt[u] = index into ovfOff (their complete lists in ovfPos), or -1; null: none
    const int* ovfAt; const long long* ovfOff; const int* ovfPos;
    const int* alphabet;        // alphabetLength per unit (or null: the host fills it in)
    // the traceback's output: op strings at the END of the units' slots (null without PATH)
    const int* opsLen; const long long* opsOff; const uint8_t* ops;
    // ---- outputs
    int* status; int* editDistance; int* numLocations; int* alphabetLength; int* alnLen;
    long long* locOff; long long* alnOff;      // [n + 1]
    int* ends; int* starts;                     // [locOff[n]]; starts may be null
    uint8_t* aln;                               // [alnOff[n]]
    long long* blockLoc; long long* blockAln;   // scratch: one entry per 256 units
};
// totals[0] = number of locations, totals[1] = number of op bytes (device memory, 2 entries)
hipError_t launch_flat_results(const FlatResultArgs& a, long long* totals, hipStream_t stream);

// edlibAlignmentToCigar (edlib.cpp:303-350) for n op strings aln[alnOff[u] .. alnOff[u + 1]).  phase 0: cigLen[u] =
// strlen + 1, cigRel / blockTot = their sums per 256 units and the bases of those groups, totals[0] = all characters;
// phase 1 (after the caller sized `out`): the NUL-terminated strings and cigOff[n + 1].
hipError_t launch_cigars(const uint8_t* aln, const long long* alnOff, int n, int standard, long long* cigLen, long long* cigRel,
                         long long* blockTot, long long* totals, char* out, long long* cigOff, int phase, hipStream_t stream);

}  // namespace edlib_amd
