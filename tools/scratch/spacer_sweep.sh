#!/bin/bash
# Sweep of tools/scratch/vmm_spacer: places, number of places, piece size.  One process per line (fresh allocator).
S=tools/scratch/vmm_spacer
run() { timeout -k 5 120 $S "$@" || exit 1; }
echo "== 48 GB, 64 MiB pieces: one place / three thirds / two thirds"
run 1000000 64 40
run 1000000 64 40 136 232
run 1000000 64 40 136
run 1000000 64 136 232
echo "== 12 GB, 64 MiB pieces, pairs of places"
run 250000 64 20
run 250000 64 20 60
run 250000 64 10 80
run 250000 64 60 110
run 250000 64 20 250
run 250000 64 100 180
run 250000 64 20 120 220
echo "== 12 GB, three thirds, piece size"
for p in 2 8 32 128 256 512 1024 2048; do run 250000 $p 40 136 232; done
echo "== 12 GB, one place, piece size"
for p in 2 64 1024; do run 250000 $p 40; done
