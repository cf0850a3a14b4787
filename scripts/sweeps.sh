#!/bin/bash
# Sweeps away from the headline configuration (search step over rows / k / batch / dimension, encoder forward over batch and
# sequence length): the place where mis-tuned thresholds show (round 2 found three).  GPU box -> gpurun_out/<tag>_sweeps.txt (usage: sweeps.sh [tag])
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03}_sweeps.txt; : > $O
kb() { python scripts/kbench.py "$@" 2>&1 | grep step | sed 's/.*step/step/' | cut -c1-75; }
echo "== search step, fp16, B=256, k=10, d=768, rows:" >> $O
for n in 1000 5000 20000 65536 300000 1000000 3000000 10000000; do echo -n "rows=$n: " >> $O; kb --n $n --steps 30 >> $O; done
echo "== k (10M rows):" >> $O
for k in 1 3 10 12 13 20 30 32 48 64 65 100; do echo -n "k=$k: " >> $O; kb --k $k --steps 15 >> $O; done
echo "== k (20000 rows):" >> $O
for k in 1 12 13 30 100 300; do echo -n "k=$k: " >> $O; kb --n 20000 --k $k --steps 30 >> $O; done
echo "== batch (10M rows):" >> $O
for b in 1 8 64 256 257 512; do echo -n "B=$b: " >> $O; kb --b $b --steps 8 >> $O; done
echo "== dimension (2M rows):" >> $O
for d in 64 128 384 768 1024; do echo -n "d=$d: " >> $O; kb --n 2000000 --d $d --steps 30 >> $O; done
echo "== index type (10M rows; fp32: 1M rows):" >> $O
echo -n "fp8: " >> $O; kb --dtype fp8 --steps 20 >> $O
echo -n "fp32 1M: " >> $O; kb --dtype fp32 --n 1000000 --steps 20 >> $O
echo "== encoder forward, PhoBERT-base shape, padded (B L):" >> $O
for shape in "1 32" "2 32" "4 32" "8 32" "10 32" "11 32" "16 32" "24 32" "32 32" "64 32" "128 32" "256 32" "128 64" "64 128" "32 256" "256 64" "64 256"; do
  python scripts/enc_bench.py $shape | grep encoder >> $O
done
cat $O
