cd $GRAFT_REPO_ROOT; O=gpurun_out/r05f; mkdir -p $O
bash tools/gzdev_variants.sh > $O/gzdev_variants.txt 2>&1; cat $O/gzdev_variants.txt | cut -c1-200
bash tools/masks_ab.sh r05f > /dev/null 2>&1; cat $O/masks_ab.txt | cut -c1-300
bash tools/exit_probe.sh > $O/exit_probe.log 2>&1; head -24 $O/exit_probe.log | cut -c1-200
bash tools/chunk_size_probe.sh r05f > /dev/null 2>&1; cat $O/chunk_size_probe.txt | cut -c1-250
