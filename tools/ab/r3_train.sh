R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3train; mkdir -p $O; cd $R
python tools/bench_train_step.py --steps 8 > $O/train_fp32.json 2>$O/train.err && cut -c1-400 $O/train_fp32.json &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/bench_train_step.py --steps 5 > $O/train_rocprof.json 2>$O/rp.err &&
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv && rm -rf $O/prof && echo DONE
