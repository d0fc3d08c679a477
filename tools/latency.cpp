// latency.cpp -- microseconds per edlibAlign() call, for any library with the edlib C ABI (dlopen by path).
//   g++ -O2 -std=c++14 tools/latency.cpp -ldl -o build/latency
//   build/latency edlib_amd/libedlib.so            (this engine: every call is a device round trip)
//   build/latency oracle/_ref/libedlib_ref.so      (the reference on one host core)
// Shapes: the reference's own published single-call numbers are 100 x 100 and 1 k x 1 k
// (/root/reference/bindings/python/README-tmpl.rst:194-215); 10 k x 10 k and a 150 bp read against 5 Mb (HW)
// are the BASELINE.json shapes seen one call at a time.  Sequences: uniform ACGT, query = target with 5 % edits.
//   build/latency <library.so> --threads N [scale]   calls per second with N host threads looping edlibAlign() concurrently
//   (the caller this serves releases the GIL around the call: /root/reference/bindings/python/edlib.pyx:128-129): the small
//   shapes only, every thread its own pair.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <string>
#include <thread>
#include <vector>
#include <cstring>

struct EqPair { char a, b; };
struct Config { int k, mode, task; const EqPair* eqs; int neq; };
struct Result { int status, ed; int* ends; int* starts; int nloc; unsigned char* aln; int alen, alpha; };
typedef Result (*align_fn)(const char*, int, const char*, int, Config);
typedef void (*free_fn)(Result);

static unsigned long long rng = 88172645463325252ull;
static unsigned next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 11); }

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: latency <library.so> [repeat-scale]\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    align_fn align = (align_fn)dlsym(h, "edlibAlign");
    free_fn release = (free_fn)dlsym(h, "edlibFreeAlignResult");
    if (!align || !release) { fprintf(stderr, "no edlibAlign in %s\n", argv[1]); return 1; }
    if (argc > 3 && !strcmp(argv[2], "--threads")) {
        const int nt = atoi(argv[3]);
        const double sc = argc > 4 ? atof(argv[4]) : 1.0;
        struct S { const char* name; int m, mode, task, reps; };
        const S small[] = {{"100 x 100 NW distance", 100, 0, 0, 2000}, {"100 x 100 NW path", 100, 0, 2, 2000},
                           {"1k x 1k NW distance", 1000, 0, 0, 1000}, {"1k x 1k NW path", 1000, 0, 2, 500}};
        printf("{\"library\": \"%s\", \"threads\": %d, \"calls_per_second\": {", argv[1], nt);
        bool first = true;
        for (const S& s : small) {
            std::vector<std::string> qs(nt), ts(nt);
            for (int k = 0; k < nt; ++k) {
                ts[k].assign(s.m, 'A');
                for (auto& c : ts[k]) c = "ACGT"[next() & 3];
                for (int i = 0; i < s.m; ++i) {
                    const unsigned r = next() % 100;
                    if (r < 2) continue;
                    if (r < 4) qs[k].push_back("ACGT"[next() & 3]);
                    qs[k].push_back(r < 5 ? "ACGT"[next() & 3] : ts[k][i]);
                }
            }
            const int reps = (int)(s.reps * sc) > 0 ? (int)(s.reps * sc) : 1;
            Config cfg{-1, s.mode, s.task, nullptr, 0};
            for (int k = 0; k < nt; ++k) { Result r = align(qs[k].data(), (int)qs[k].size(), ts[k].data(), s.m, cfg); release(r); }
            std::vector<long long> chk(nt, 0);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int k = 0; k < nt; ++k) th.emplace_back([&, k] {
                for (int i = 0; i < reps; ++i) { Result r = align(qs[k].data(), (int)qs[k].size(), ts[k].data(), s.m, cfg); chk[k] += r.ed + r.status * 1000000; release(r); }
            });
            for (auto& t : th) t.join();
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%s\"%s\": %.0f", first ? "" : ", ", s.name, (double)reps * nt / sec);
            first = false;
        }
        printf("}}\n");
        return 0;
    }
    const double scale = argc > 2 ? atof(argv[2]) : 1.0;
    struct Shape { const char* name; int m, T, mode, task, reps; };
    const Shape shapes[] = {
        {"100 x 100 NW distance", 100, 100, 0, 0, 2000}, {"100 x 100 NW path", 100, 100, 0, 2, 2000},
        {"1k x 1k NW distance", 1000, 1000, 0, 0, 1000}, {"1k x 1k NW path", 1000, 1000, 0, 2, 500},
        {"10k x 10k NW distance", 10000, 10000, 0, 0, 200}, {"10k x 10k NW path", 10000, 10000, 0, 2, 100},
        {"100k x 100k NW distance", 100000, 100000, 0, 0, 20}, {"150 x 5Mb HW distance", 150, 5000000, 2, 0, 20},
        {"150 x 5Mb HW path", 150, 5000000, 2, 2, 20},
    };
    printf("{\"library\": \"%s\", \"us_per_call\": {", argv[1]);
    bool first = true;
    for (const Shape& s : shapes) {
        std::string t(s.T, 'A'), q;
        for (auto& c : t) c = "ACGT"[next() & 3];
        const int from = s.T > s.m ? (int)(next() % (unsigned)(s.T - s.m)) : 0;
        for (int i = 0; i < s.m; ++i) {
            const unsigned r = next() % 100;
            if (r < 2) continue;                                   // deletion
            if (r < 4) q.push_back("ACGT"[next() & 3]);            // insertion
            q.push_back(r < 5 ? "ACGT"[next() & 3] : t[from + i]); // substitution / copy
        }
        Config cfg{-1, s.mode, s.task, nullptr, 0};
        const int reps = (int)(s.reps * scale) > 0 ? (int)(s.reps * scale) : 1;
        long long check = 0;
        for (int w = 0; w < 3; ++w) { Result r = align(q.data(), (int)q.size(), t.data(), s.T, cfg); check += r.ed; release(r); }
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) { Result r = align(q.data(), (int)q.size(), t.data(), s.T, cfg); check += r.ed + r.status * 1000000; release(r); }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("%s\"%s\": %.2f", first ? "" : ", ", s.name, us);
        first = false;
        if (check < 0) printf(" ");
    }
    printf("}}\n");
    return 0;
}
