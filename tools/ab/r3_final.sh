# Round-3 evidence passes on one MI355X: bench line, kernel-trace stats, PMC traffic passes (separate runs), SQ pass.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3fin; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err && cut -c1-300 $O/bench.json &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-budget 0 --median-steps 0 --streams 1 --lanes 1 > $O/under_rocprof.json 2>$O/kt.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --median-steps 0 --streams 1 --lanes 1 > $O/pf.json 2>$O/pf.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --median-steps 0 --streams 1 --lanes 1 > $O/pw.json 2>$O/pw.err &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/psq -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --median-steps 0 --streams 1 --lanes 1 > $O/psq.json 2>$O/psq.err &&
cd $R && F=$(ls $O/pf/*/*counter_collection.csv | head -1) && W=$(ls $O/pw/*/*counter_collection.csv | head -1) && S=$(ls $O/psq/*/*counter_collection.csv | head -1) &&
python tools/pmc_traffic.py $F $W $O/pmc_traffic.json &&
gzip -c $F > $O/pmc_FETCH_SIZE_counter_collection.csv.gz && gzip -c $W > $O/pmc_WRITE_SIZE_counter_collection.csv.gz && gzip -c $S > $O/pmc_SQ_counter_collection.csv.gz &&
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv && rm -rf $O/kt $O/pf $O/pw $O/psq && ls -la $O && echo ALLDONE
