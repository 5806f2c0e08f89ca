mkdir -p gpurun_out/r06
O=gpurun_out/r06/f_front_variants4.txt
: > $O
run() { echo "## $*" >> $O; env "$@" SWEEP_CHECK=${CHECK:-0} timeout 300 python tools/bait_sweep.py 33333334 $SIZES >> $O 2>&1; }
CHECK=1 SIZES=16569,25000,33000,50000,70000,100000 run X=default
SIZES=25000,33000,50000,70000 run SWEEP_OPTS=front=1
SIZES=50000,70000,100000 run SWEEP_OPTS=front=3
echo "## variants of stage 1 (base: two table blocks asked at a time; b4 / b8: four / eight)" >> $O
bash tools/ab_variants.sh "base b4 b8" "31:16569,21:16569,25:16569,41:16569" >> $O 2>&1
cat $O
