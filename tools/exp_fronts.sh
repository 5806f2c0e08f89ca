mkdir -p gpurun_out/r06
O=gpurun_out/r06/e_front_variants3.txt
: > $O
run() { echo "## $*" >> $O; env "$@" SWEEP_CHECK=${CHECK:-0} timeout 300 python tools/bait_sweep.py 33333334 $SIZES >> $O 2>&1; }
CHECK=1 SIZES=16569,25000,33000,50000,100000,120000,350000,1000000 run X=default
SIZES=25000,33000,50000 run SWEEP_OPTS=front=0
cat $O
