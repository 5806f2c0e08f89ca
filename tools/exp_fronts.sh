mkdir -p gpurun_out/r06
O=gpurun_out/r06/k_mode4.txt
: > $O
run() { echo "## $*" >> $O; env "$@" SWEEP_CHECK=${CHECK:-0} timeout 400 python tools/bait_sweep.py 33333334 $SIZES >> $O 2>&1; }
CHECK=1 SIZES=100000,150000,200000,350000,500000,700000,1000000 run X=default
SIZES=60000,100000,700000,1000000 run SWEEP_OPTS=front=4
SIZES=200000,350000,500000 run SWEEP_OPTS=front=2
cat $O
