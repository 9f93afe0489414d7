set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3v2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
c=WRITE_SIZE
timeout -k 10 800 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sequence-leg > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
python3 $R/profiles/pmc_sum.py k_sor_stream $O/pmc_$c > $O/pmc_$c.txt; python3 $R/profiles/pmc_sum.py k_sor_fused $O/pmc_$c >> $O/pmc_$c.txt; python3 $R/profiles/pmc_sum.py k_peac_grow $O/pmc_$c >> $O/pmc_$c.txt; rm -rf $O/pmc_$c
echo write-done
