#!/bin/bash
# Compile the reference's OWN clients (test driver, CLI, hello-world), unmodified and from where
# they lie under $REF, against THIS repository's libedlib.so.  Outputs go to oracle/_ref/ (git-ignored,
# travels to the GPU box).  This is the drop-in check of INTEGRATION.md §1.
set -e
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -f "$REF/test/runTests.cpp" ] || { echo "no $REF: keeping prebuilt clients"; exit 0; }
[ -f "$ROOT/edlib_amd/libedlib.so" ] || { echo "build edlib_amd/libedlib.so first"; exit 1; }
mkdir -p "$ROOT/oracle/_ref"
LINK="-L$ROOT/edlib_amd -l:libedlib.so -Wl,-rpath,\$ORIGIN/../../edlib_amd"
g++ -O2 -std=c++14 -I"$REF/edlib/include" -I"$REF/test" "$REF/test/runTests.cpp" $LINK -o "$ROOT/oracle/_ref/runTests_amd"
g++ -O2 -std=c++14 -I"$REF/edlib/include" "$REF/apps/aligner/aligner.cpp" $LINK -o "$ROOT/oracle/_ref/aligner_amd"
gcc -O2 -I"$REF/edlib/include" "$REF/apps/hello-world/helloWorld.c" $LINK -o "$ROOT/oracle/_ref/hello_amd"
# the pure reference CLI (reference library underneath): expected output for the CLI tests
g++ -O2 -std=c++14 -I"$REF/edlib/include" "$REF/apps/aligner/aligner.cpp" "$REF/edlib/src/edlib.cpp" -o "$ROOT/oracle/_ref/aligner_ref"
install -m 644 "$REF/apps/aligner/test_data/query.fasta" "$ROOT/oracle/_ref/aligner_query.fasta"
install -m 644 "$REF/apps/aligner/test_data/target.fasta" "$ROOT/oracle/_ref/aligner_target.fasta"
echo "built oracle/_ref/{runTests_amd,aligner_amd,hello_amd}"
