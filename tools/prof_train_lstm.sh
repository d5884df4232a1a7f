R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in new old; do
  if [ $L = old ]; then export AMTX_LIB_PATH=$R/tools/_dbg/libamtx_oldlstm.so; else unset AMTX_LIB_PATH; fi
  rm -rf $O/tr_$L
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr_$L -o prof -- python3 $R/tools/bench_train.py > $O/tr_$L.log 2>&1
  DB=$(ls $O/tr_$L/*.db $O/tr_$L/*/*.db 2>/dev/null | head -1)
  cd $R; echo "== $L"; python3 tools/rocpd_summary.py $DB | grep -i "bilstm" | cut -c1-150; cd /tmp
  rm -rf $O/tr_$L
done
