#!/bin/bash
# Host-side memory check (AddressSanitizer + UBSan; the GPU pool has no sanitizer support, so this runs on the CPU build):
# libswsharp (the C++ host API), the test-only oracle stand-in for libs4g_hip.so and the reference's unmodified CLI
# (compiled in place from /root/reference) are built with -fsanitize=address,undefined under build/asan/ and the CLI is
# run on the reference's fixture; its three outputs must equal tests/golden/dropin_predictions/ and the sanitizers must
# stay silent (leak check included).  Usage: tests/cpu_shim/asan_check.sh [thread]   (needs /root/reference; ~3 min)
# With "thread" the same programs are built with -fsanitize=thread under build/tsan/ and the CLI runs with -t 4; reports whose
# innermost frames lie in the reference's own sources (its lgamma/signgam use in sift_scores.cpp) are not ours and are ignored.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"
REF=${REF:-/root/reference}
MODE=${1:-address}
if [ "$MODE" = thread ]; then B=build/tsan; SAN="-fsanitize=thread -g -O1"; else B=build/asan; SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"; fi
mkdir -p $B/lib $B/out
gcc $SAN -std=c11 -fPIC -c oracle/sw_oracle.c -o $B/sw_oracle.o
gcc $SAN -std=c11 -fPIC -c oracle/kmer_prefilter.c -o $B/kmer_prefilter.o
g++ $SAN -std=c++17 -fPIC -shared -Iinclude -o $B/lib/libs4g_hip.so tests/cpu_shim/s4g_oracle_shim.cpp $B/sw_oracle.o $B/kmer_prefilter.o -lpthread
g++ $SAN -std=c++17 -fPIC -shared -Iinclude -o $B/lib/libswsharp.so sift4g_amd/csrc/swsharp/core.cpp sift4g_amd/csrc/swsharp/evalue.cpp \
    sift4g_amd/csrc/swsharp/align.cpp sift4g_amd/csrc/swsharp/output.cpp -L$B/lib -ls4g_hip -lpthread
g++ $SAN -std=c++11 -w -Iinclude -o $B/sift4g $REF/sift4g/src/*.cpp -L$B/lib -lswsharp -ls4g_hip -lpthread -lm
g++ $SAN -std=c++11 -w -Iinclude -I$REF/sift4g/src -o $B/sift4g_batched $(ls $REF/sift4g/src/*.cpp | grep -v /database_alignment.cpp) \
    sift4g_amd/csrc/dropin/batched_database_alignment.cpp -L$B/lib -lswsharp -ls4g_hip -lpthread -lm
g++ $SAN -std=c++11 -w -Iinclude -I$REF/sift4g/src -o $B/sift4g_devsearch $(ls $REF/sift4g/src/*.cpp | grep -v "/database_alignment.cpp\|/database_search.cpp\|/hash.cpp") \
    sift4g_amd/csrc/dropin/batched_database_alignment.cpp sift4g_amd/csrc/dropin/device_database_search.cpp -L$B/lib -lswsharp -ls4g_hip -lpthread -lm
g++ $SAN -std=c++17 -Iinclude -o $B/shotgun_demo tools/shotgun_demo.cpp -L$B/lib -lswsharp -ls4g_hip -lpthread
F=tests/golden/test_files
rm -rf $B/out/*
LD_LIBRARY_PATH=$B/lib $B/sift4g -q $F/query.fasta --subst $F/ -d $F/sample_protein_database.fa --cards 0 --out $B/out --sub-results -t 4 \
    > $B/out/stdout.txt 2> $B/out/stderr.txt || [ "$MODE" = thread ]
for f in LACI_ECOLI.SIFTprediction PURR_SALTY.SIFTprediction alignments.txt; do cmp $B/out/$f tests/golden/dropin_predictions/$f; done
# the batched wrapper, database read in several parts
mkdir -p $B/out/batched
S4G_CHUNK_BYTES=1500 LD_LIBRARY_PATH=$B/lib $B/sift4g_batched -q $F/query.fasta --subst $F/ -d $F/sample_protein_database.fa --cards 0 \
    --out $B/out/batched --sub-results -t 4 > $B/out/batched/stdout.txt 2>> $B/out/stderr.txt || [ "$MODE" = thread ]
for f in LACI_ECOLI.SIFTprediction PURR_SALTY.SIFTprediction alignments.txt; do cmp $B/out/batched/$f tests/golden/dropin_predictions/$f; done
# the device-prefilter build (flat FASTA reader, parts handed over on helper threads), both stages in several parts
mkdir -p $B/out/devsearch
S4G_CHUNK_BYTES=1500 S4G_PREFILTER_CHUNK_BYTES=1200 LD_LIBRARY_PATH=$B/lib $B/sift4g_devsearch -q $F/query.fasta --subst $F/ -d $F/sample_protein_database.fa \
    --cards 0 --out $B/out/devsearch --sub-results -t 4 > $B/out/devsearch/stdout.txt 2>> $B/out/stderr.txt || [ "$MODE" = thread ]
for f in LACI_ECOLI.SIFTprediction PURR_SALTY.SIFTprediction alignments.txt; do cmp $B/out/devsearch/$f tests/golden/dropin_predictions/$f; done
# the same build with TWO cards: one host thread per card works on the same flat database (the placeholders' real chains are
# built on first use from either thread, under the database's lock: ADVICE r3)
mkdir -p $B/out/cards00
S4G_CHUNK_BYTES=1500 S4G_PREFILTER_CHUNK_BYTES=1200 LD_LIBRARY_PATH=$B/lib $B/sift4g_devsearch -q $F/query.fasta --subst $F/ -d $F/sample_protein_database.fa \
    --cards 00 --out $B/out/cards00 --sub-results -t 4 > $B/out/cards00/stdout.txt 2>> $B/out/stderr.txt || [ "$MODE" = thread ]
for f in LACI_ECOLI.SIFTprediction PURR_SALTY.SIFTprediction alignments.txt; do cmp $B/out/cards00/$f tests/golden/dropin_predictions/$f; done
for ev in 0 1; do      # the batched entry points: callback selection and e-value driven selection
    LD_LIBRARY_PATH=$B/lib $B/shotgun_demo $F/query.fasta $F/sample_protein_database.fa 400 1e-4 4 $ev > $B/out/shotgun_$ev.txt 2>> $B/out/stderr.txt
done
if [ "$MODE" = thread ]; then
    # a report is ours when libswsharp / the stand-in appears above the pool's own dispatch frames
    if grep -A3 "WARNING: ThreadSanitizer" $B/out/stderr.txt | grep -q "swsharp/\|s4g_oracle_shim"; then grep -m5 -A6 "WARNING: ThreadSanitizer" $B/out/stderr.txt; exit 1; fi
elif grep -q "Sanitizer\|runtime error" $B/out/stderr.txt; then grep -m5 "Sanitizer\|runtime error" $B/out/stderr.txt; exit 1; fi
echo "sanitizer check ($MODE): clean, outputs identical to tests/golden/dropin_predictions"
