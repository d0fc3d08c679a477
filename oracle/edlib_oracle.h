/*
 * oracle/edlib_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C99 restatement of the edit-distance hot path of Martinsos/edlib
 * (reference: edlib/src/edlib.cpp, v1.2.6).  It exists so that the HIP engine
 * in edlib_amd/ can be checked bit-for-bit on a machine that has no copy of
 * the reference sources (the GPU box).  Nothing under edlib_amd/ may include,
 * link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 *   (a) the reference's own known-answer tests (test/runTests.cpp:269-587,
 *       bindings/python/test.py:6-80), re-typed in tests/golden/kat.json,
 *   (b) tests/golden/fuzz_ref.json -- outputs of the *compiled reference*
 *       (oracle/_ref/libedlib_ref.so, built by oracle/Makefile from
 *       /root/reference/edlib/src/edlib.cpp) on seeded random inputs, produced
 *       by oracle/gen_golden.py, and
 *   (c) when oracle/_ref/libedlib_ref.so is present, a live differential fuzz.
 */
#ifndef EDLIB_ORACLE_H
#define EDLIB_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Same numeric values as the reference enums (edlib/include/edlib.h:36-81). */
enum { ORACLE_MODE_NW = 0, ORACLE_MODE_SHW = 1, ORACLE_MODE_HW = 2 };
enum { ORACLE_TASK_DISTANCE = 0, ORACLE_TASK_LOC = 1, ORACLE_TASK_PATH = 2 };
enum { ORACLE_CIGAR_STANDARD = 0, ORACLE_CIGAR_EXTENDED = 1 };
enum { ORACLE_OK = 0, ORACLE_ERROR = 1, ORACLE_UNSUPPORTED = 2 };

typedef struct { char first; char second; } OracleEqualityPair;

/* Field-for-field the layout of EdlibAlignResult (edlib.h:162-218). */
typedef struct {
    int status;
    int editDistance;
    int* endLocations;
    int* startLocations;
    int numLocations;
    unsigned char* alignment;
    int alignmentLength;
    int alphabetLength;
} OracleAlignResult;

/* Restates edlibAlign (edlib.cpp:146-301).  status == ORACLE_UNSUPPORTED is
 * returned for the one regime this oracle does not restate: TASK_PATH where
 * the reference would switch to Hirschberg (edlib.cpp:1188-1211). */
OracleAlignResult oracle_align(const char* query, int queryLength,
                               const char* target, int targetLength,
                               int k, int mode, int task,
                               const OracleEqualityPair* eq, int numEq);

void oracle_free_result(OracleAlignResult* r);

/* Restates edlibAlignmentToCigar (edlib.cpp:303-350). malloc'd, NUL-terminated. */
char* oracle_cigar(const unsigned char* alignment, int alignmentLength, int format);

/* Textbook O(m*T) dynamic programme, independent of the bit-vector code; used
 * to cross-check the restatement itself (mirrors test/SimpleEditDistance.h:24-106
 * in purpose, written independently). Positions are malloc'd. */
int oracle_simple_dp(const unsigned char* query, int queryLength,
                     const unsigned char* target, int targetLength,
                     int mode, int* score, int** positions, int* numPositions);

#ifdef __cplusplus
}
#endif
#endif
