R=$PWD; cd /tmp; export TMPDIR=/tmp
for n in 1000000 16384 1024; do rm -rf /tmp/pm; PER_N=$n timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o t -- python $R/tools/per_bench.py > /tmp/pm.log 2>&1; echo "N=$n $(grep k_per_sample /tmp/pm/t_kernel_stats.csv | cut -d, -f1-4)"; done
