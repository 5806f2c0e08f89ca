mkdir -p gpurun_out/r06
O=gpurun_out/r06/g_front_k21.txt
: > $O
run() { echo "## $*" >> $O; env "$@" SWEEP_CHECK=0 timeout 300 python tools/bait_sweep.py 33333334 $SIZES 21 >> $O 2>&1; }
SIZES=16569,33000,50000,100000,350000 run X=default
SIZES=33000,50000,100000 run SWEEP_OPTS=front=0
SIZES=50000,100000 run SWEEP_OPTS=front=2
cat $O
