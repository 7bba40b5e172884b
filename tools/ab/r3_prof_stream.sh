R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3pstream; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/prof_stream.py > $O/out.txt 2>$O/err.txt &&
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv && rm -rf $O/kt && cat $O/out.txt
