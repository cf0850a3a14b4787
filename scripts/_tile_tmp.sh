cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tilemat; rm -rf $O; mkdir -p $O
for t in -1 0 1 2 3 4; do
  if [ $t = -1 ]; then unset VQA_GEMM_TILE; else export VQA_GEMM_TILE=$t; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$t -- python3 $R/scripts/enc_bench.py 256 32 > $O/t$t.log 2>&1
  echo "== tile $t: $(grep encoder $O/t$t.log)"
  f=$(ls $O/t$t/*/*kernel_stats.csv | head -1)
  grep gemm_tile $f | cut -d, -f1,2,4 | sed 's/_ZN12_GLOBAL__N_116gemm_tile_kernelI//; s/EEvPKDF16.*iiiiii//'
  cp $(ls $O/t$t/*/*kernel_trace.csv | head -1) $O/trace_t$t.csv
  rm -rf $O/t$t
done
