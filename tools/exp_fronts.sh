mkdir -p gpurun_out/r06
O=gpurun_out/r06/i_mode3_at_16k.txt
: > $O
run() { echo "## $*" >> $O; env "$@" SWEEP_CHECK=0 timeout 300 python tools/bait_sweep.py 33333334 $SIZES >> $O 2>&1; }
for rep in 1 2 3; do
SIZES=16569 run X=default
SIZES=16569 run SWEEP_OPTS=front=3
done
cat $O
