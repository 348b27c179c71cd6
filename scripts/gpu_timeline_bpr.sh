set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace -d $OUT/prof_tl -o tl -- python $GRAFT_REPO_ROOT/scripts/gpu_probe_bpr_epochs.py 16 > $OUT/r05_l_epochs.txt 2>&1
DB=$(find $OUT/prof_tl -name '*_results.db' | head -1)
python $GRAFT_REPO_ROOT/scripts/rocpd_timeline.py $DB 70 > $OUT/r05_l_timeline_bpr_d16.txt 2>&1
rm -rf $OUT/prof_tl
tail -5 $OUT/r05_l_epochs.txt; tail -50 $OUT/r05_l_timeline_bpr_d16.txt
