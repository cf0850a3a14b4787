cd $GRAFT_REPO_ROOT
for n in 500000 1000000 2000000 3000000; do for sm in 0 4 8 16 40; do
echo -n "n=$n stage_min=$sm: "; VQA_STAGE_MIN=$sm python scripts/kbench.py --n $n --steps 40 2>&1 | grep step | sed 's/.*step/step/' | cut -c1-42
done; done
