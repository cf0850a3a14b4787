cd $GRAFT_REPO_ROOT
for shape in "256 32" "128 64" "64 128" "32 256" "256 64" "128 128" "64 256" "2 128" "4 128" "1 256" "3 100"; do
python scripts/enc_bench.py $shape | grep encoder | cut -c1-72
done
for shape in "256 64" "128 128"; do
ENC_PACK=1 python scripts/enc_bench.py $shape | grep encoder | cut -c1-90
done
