/*
 * edlib.h -- drop-in C boundary of the MI355X edit-distance engine.
 *
 * This header declares, with the same names, field order, enum values and
 * calling convention, the public interface of Martinsos/edlib v1.2.6
 * (reference: edlib/include/edlib.h:1-277).  A program or binding compiled
 * against the reference header links against libedlib.so built from this
 * repository without change; only the implementation underneath is new
 * (hand-written gfx950 HIP kernels).
 *
 * ABI notes (x86-64 SysV, checked by tests/test_abi.py):
 *   sizeof(EdlibAlignConfig) == 32, passed by value;
 *   sizeof(EdlibAlignResult) == 48, returned by value;
 *   result arrays are libc malloc() memory owned by the caller
 *   (edlibFreeAlignResult() or plain free()).
 *
 * Behaviour that differs from the reference:
 *   - when no MI355X-class device / HIP runtime is usable, edlibAlign() does
 *     NOT fall back to a CPU path: it returns status == EDLIB_STATUS_ERROR
 *     and prints one line to stderr.
 */
#ifndef EDLIB_H
#define EDLIB_H

/* reference edlib.h:10-23 -- symbol visibility */
#ifdef EDLIB_SHARED
#  ifdef _WIN32
#    ifdef EDLIB_BUILD
#      define EDLIB_API __declspec(dllexport)
#    else
#      define EDLIB_API __declspec(dllimport)
#    endif
#  else
#    define EDLIB_API __attribute__((visibility("default")))
#  endif
#else
#  define EDLIB_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* reference edlib.h:29-31 */
#define EDLIB_STATUS_OK 0
#define EDLIB_STATUS_ERROR 1

/* reference edlib.h:36-62.  Which gaps are free:
 *   NW  global  -- none;
 *   SHW prefix  -- target characters after the query end;
 *   HW  infix   -- target characters before the query start and after its end. */
typedef enum {
    EDLIB_MODE_NW,
    EDLIB_MODE_SHW,
    EDLIB_MODE_HW
} EdlibAlignMode;

/* reference edlib.h:67-71.  How much to compute. */
typedef enum {
    EDLIB_TASK_DISTANCE,   /* distance + end locations                    */
    EDLIB_TASK_LOC,        /* ... + start locations                       */
    EDLIB_TASK_PATH        /* ... + alignment of the first (start,end)    */
} EdlibAlignTask;

/* reference edlib.h:78-81 */
typedef enum {
    EDLIB_CIGAR_STANDARD,  /* M I D   (match and mismatch both 'M')       */
    EDLIB_CIGAR_EXTENDED   /* = I D X                                     */
} EdlibCigarFormat;

/* reference edlib.h:84-87 -- codes stored in EdlibAlignResult.alignment */
#define EDLIB_EDOP_MATCH 0
#define EDLIB_EDOP_INSERT 1      /* consumes a query character only  */
#define EDLIB_EDOP_DELETE 2      /* consumes a target character only */
#define EDLIB_EDOP_MISMATCH 3

/* reference edlib.h:92-95 -- declares two characters equal */
typedef struct {
    char first;
    char second;
} EdlibEqualityPair;

/* reference edlib.h:100-140 */
typedef struct {
    int k;                    /* >= 0: report -1 if the distance exceeds k; < 0: no bound */
    EdlibAlignMode mode;
    EdlibAlignTask task;
    const EdlibEqualityPair* additionalEqualities;   /* may be NULL */
    int additionalEqualitiesLength;
} EdlibAlignConfig;

/* reference edlib.h:146-150 */
EDLIB_API EdlibAlignConfig edlibNewAlignConfig(
    int k, EdlibAlignMode mode, EdlibAlignTask task,
    const EdlibEqualityPair* additionalEqualities,
    int additionalEqualitiesLength
);

/* reference edlib.h:156 -- k = -1, NW, DISTANCE, no extra equalities */
EDLIB_API EdlibAlignConfig edlibDefaultAlignConfig(void);

/* reference edlib.h:162-218 */
typedef struct {
    int status;               /* EDLIB_STATUS_OK / EDLIB_STATUS_ERROR                     */
    int editDistance;         /* -1 when k >= 0 and the distance is larger than k        */
    int* endLocations;        /* 0-based end positions in target, ascending; NULL if none */
    int* startLocations;      /* matching start positions (LOC / PATH), else NULL         */
    int numLocations;
    unsigned char* alignment; /* EDLIB_EDOP_* codes for the first location (PATH)         */
    int alignmentLength;
    int alphabetLength;       /* distinct byte values in query and target together       */
} EdlibAlignResult;

/* reference edlib.h:224 */
EDLIB_API void edlibFreeAlignResult(EdlibAlignResult result);

/* reference edlib.h:242-246.  Sequences are raw bytes (not NUL terminated). */
EDLIB_API EdlibAlignResult edlibAlign(
    const char* query, int queryLength,
    const char* target, int targetLength,
    const EdlibAlignConfig config
);

/* reference edlib.h:268-271.  malloc'd NUL-terminated string; NULL on a bad
 * format or an op code > 3.  Caller free()s. */
EDLIB_API char* edlibAlignmentToCigar(
    const unsigned char* alignment, int alignmentLength,
    EdlibCigarFormat cigarFormat
);

#ifdef __cplusplus
}
#endif

#endif /* EDLIB_H */
