cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_encoder.py -x -q 2>&1 | tail -3
O=$GRAFT_REPO_ROOT/gpurun_out/att; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/d.log 2>&1
grep encoder $O/d.log; grep attention $(ls $O/d/*/*kernel_stats.csv | head -1) | cut -d, -f1-4 | cut -c1-120
export ENC_PACK=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/p.log 2>&1
grep encoder $O/p.log; grep attention $(ls $O/p/*/*kernel_stats.csv | head -1) | cut -d, -f1-4 | cut -c1-120
unset ENC_PACK
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 64 128 > $O/l.log 2>&1
grep encoder $O/l.log; grep attention $(ls $O/l/*/*kernel_stats.csv | head -1) | cut -d, -f1-4 | cut -c1-120
rm -rf $O
