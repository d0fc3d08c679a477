// edlib-aligner-batch -- batch-aware front end with the command line and the output of the
// reference's CLI (apps/aligner/aligner.cpp:49-60 flags, :238-257 / :200-221 output), but ONE device
// batch instead of one edlibAlign() per query (the loop at aligner.cpp:162-225).
//
//   edlib-aligner-batch [-m HW|NW|SHW] [-n N] [-k K] [-p] [-l] [-f NICE|CIG_STD|CIG_EXT] [-s] [-r R]
//                       <queries.fasta> <target.fasta>
//
// The reference tightens k after every query when -n N is given (aligner.cpp:183-195), which makes
// query i's result depend on queries 0..i-1.  A result for a smaller k is the same result or "-1",
// so the batch is computed once with the loosest k and the tightening is replayed on the host.
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

#include "edlib.h"
#include "edlib_amd.h"

// FASTA reader with the reference's conventions (aligner.cpp:290-328): header lines start with '>',
// CR/LF dropped, case preserved, a file without any header is one sequence.
static bool read_fasta(const char* path, std::vector<std::string>& seqs) {
    FILE* f = fopen(path, "r");
    if (!f) return false;
    bool header = false, open = false;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) {
        for (size_t i = 0; i < n; ++i) {
            const char c = buf[i];
            if (header) { if (c == '\n') header = false; continue; }
            if (c == '>') { header = true; open = false; continue; }
            if (c == '\r' || c == '\n') continue;
            if (!open) { seqs.emplace_back(); open = true; }
            seqs.back().push_back(c);
        }
    }
    fclose(f);
    return true;
}

// aligner.cpp:331-377, same layout: 50 columns per row, T / match / Q lines with index ranges
static void print_nice(const char* q, const char* t, const unsigned char* aln, int len, int endPos, int mode) {
    int ti = -1, qi = -1;
    if (mode == EDLIB_MODE_HW) {
        ti = endPos;
        for (int i = 0; i < len; ++i) if (aln[i] != EDLIB_EDOP_INSERT) --ti;
    }
    for (int s = 0; s < len; s += 50) {
        const int e = std::min(len, s + 50);
        printf("T: ");
        int t0 = -1;
        for (int j = s; j < e; ++j) {
            if (aln[j] == EDLIB_EDOP_INSERT) printf("-"); else printf("%c", t[++ti]);
            if (j == s) t0 = ti;
        }
        printf(" (%d - %d)\n   ", std::max(t0, 0), ti);
        for (int j = s; j < e; ++j) printf(aln[j] == EDLIB_EDOP_MATCH ? "|" : " ");
        printf("\nQ: ");
        int q0 = qi;
        for (int j = s; j < e; ++j) {
            if (aln[j] == EDLIB_EDOP_DELETE) printf("-"); else printf("%c", q[++qi]);
            if (j == s) q0 = qi;
        }
        printf(" (%d - %d)\n\n", std::max(q0, 0), qi);
    }
}

int main(int argc, char* const argv[]) {
    bool silent = false, wantPath = false, wantStarts = false, bad = false;
    std::string mode = "NW", fmt = "NICE";
    int bestN = 0, kArg = -1, repeats = 1, opt;
    while ((opt = getopt(argc, argv, "m:n:k:f:r:spl")) >= 0) {
        switch (opt) {
            case 'm': mode = optarg; break;
            case 'n': bestN = atoi(optarg); break;
            case 'k': kArg = atoi(optarg); break;
            case 'f': fmt = optarg; break;
            case 's': silent = true; break;
            case 'p': wantPath = true; break;
            case 'l': wantStarts = true; break;
            case 'r': repeats = atoi(optarg); break;
            default: bad = true;
        }
    }
    if (optind + 2 != argc || bad) {
        fprintf(stderr, "Usage: %s [-s] [-m HW|NW|SHW] [-n N] [-k K] [-p] [-l] [-f NICE|CIG_STD|CIG_EXT] [-r N] "
                        "<queries.fasta> <target.fasta>\n", argv[0]);
        return 1;
    }
    if (fmt != "NICE" && fmt != "CIG_STD" && fmt != "CIG_EXT") { printf("Invalid alignment path format (-f)!\n"); return 1; }
    EdlibAlignMode modeCode;
    if (mode == "SHW") modeCode = EDLIB_MODE_SHW; else if (mode == "HW") modeCode = EDLIB_MODE_HW;
    else if (mode == "NW") modeCode = EDLIB_MODE_NW; else { printf("Invalid mode (-m)!\n"); return 1; }
    printf("Using %s alignment mode.\n", mode.c_str());
    const EdlibAlignTask task = wantPath ? EDLIB_TASK_PATH : wantStarts ? EDLIB_TASK_LOC : EDLIB_TASK_DISTANCE;

    std::vector<std::string> queries, targets;
    printf("Reading queries...\n");
    if (!read_fasta(argv[optind], queries)) { printf("Error: There is no file with name %s\n", argv[optind]); return 1; }
    long long residues = 0;
    for (auto& q : queries) residues += (long long)q.size();
    printf("Read %d queries, %lld residues total.\n", (int)queries.size(), residues);
    printf("Reading target fasta file...\n");
    if (!read_fasta(argv[optind + 1], targets) || targets.empty()) { printf("Error: There is no file with name %s\n", argv[optind + 1]); return 1; }
    const std::string& target = targets[0];                       // first record only (aligner.cpp:142-143)
    printf("Read target, %d residues.\n", (int)target.size());

    printf("\nComparing queries to target...\n");
    const int n = (int)queries.size();
    std::vector<const char*> qptr(n); std::vector<int> qlen(n);
    for (int i = 0; i < n; ++i) { qptr[i] = queries[i].data(); qlen[i] = (int)queries[i].size(); }
    std::vector<EdlibAlignResult> res(n);
    const auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < repeats; ++rep) {
        if (rep) for (auto& r : res) edlibFreeAlignResult(r);
        if (edlibAlignBatchSharedTarget(qptr.data(), qlen.data(), n, target.data(), (int)target.size(),
                                        edlibNewAlignConfig(kArg, modeCode, task, NULL, 0), res.data()) != EDLIB_STATUS_OK) {
            fprintf(stderr, "edlib-aligner-batch: %s\n", edlibAmdLastError());
            return 1;
        }
    }
    // replay of the reference's best-N tightening of k (aligner.cpp:183-195)
    std::priority_queue<int> bestScores;
    int k = kArg;
    for (int i = 0; i < n; ++i) {
        EdlibAlignResult& r = res[i];
        if (k >= 0 && r.editDistance > k) {                       // what edlibAlign(k) would have said
            edlibFreeAlignResult(r);
            r.editDistance = -1; r.endLocations = r.startLocations = NULL; r.numLocations = 0;
            r.alignment = NULL; r.alignmentLength = 0;
        }
        if (bestN > 0 && r.editDistance >= 0) {
            bestScores.push(r.editDistance);
            if ((int)bestScores.size() > bestN) bestScores.pop();
            if ((int)bestScores.size() == bestN) { k = bestScores.top() - 1; if (kArg >= 0 && kArg < k) k = kArg; }
        }
        if (wantPath && !silent && r.alignment) {                 // aligner.cpp:200-221
            printf("\nQuery #%d (%d residues): score = %d\n", i, qlen[i], r.editDistance);
            if (fmt == "NICE") print_nice(qptr[i], target.data(), r.alignment, r.alignmentLength, r.endLocations[0], modeCode);
            else {
                printf("Cigar:\n");
                char* c = edlibAlignmentToCigar(r.alignment, r.alignmentLength, fmt == "CIG_STD" ? EDLIB_CIGAR_STANDARD : EDLIB_CIGAR_EXTENDED);
                if (c) { printf("%s\n", c); free(c); } else printf("Error while printing cigar!\n");
            }
        }
    }
    if (!silent && !wantPath) {                                   // aligner.cpp:228-258
        int limit = -1;
        printf("\n");
        if (!bestScores.empty()) { printf("%d best scores:\n", (int)bestScores.size()); limit = bestScores.top(); }
        else printf("Scores:\n");
        printf("<query number>: <score>, <num_locations>, [(<start_location_in_target>, <end_location_in_target>)]\n");
        for (int i = 0; i < n; ++i) {
            const EdlibAlignResult& r = res[i];
            if (r.editDistance < 0 || (limit != -1 && r.editDistance > limit)) continue;
            printf("#%d: %d  %d", i, r.editDistance, r.numLocations);
            if (r.numLocations > 0) {
                printf("  [");
                for (int j = 0; j < r.numLocations; ++j) {
                    printf(" (");
                    if (r.startLocations) printf("%d", r.startLocations[j]); else printf("?");
                    printf(", %d)", r.endLocations[j]);
                }
                printf(" ]");
            }
            printf("\n");
        }
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("\nWall time of searching: %lf\n", secs);
    for (auto& r : res) edlibFreeAlignResult(r);
    return 0;
}
