#!/bin/bash
# round 5: does the file-fed rate depend on which NUMA node holds the page cache of the capture and runs the readers?  (GPU on node 0: CPUs 0-63,128-191)
out=gpurun_out/r05io2; mkdir -p $out
cat /sys/class/drm/card*/device/numa_node | tr '\n' ' ' >> $out/io.txt; lscpu | grep -i "numa node" >> $out/io.txt
for cpus in "" "0-63" "64-127"; do
  echo "## TWX_IO_CPUS=$cpus (writer of the tmpfs file and all reader threads)" >> $out/io.txt
  TWX_IO_CPUS=$cpus python tools/io_rate.py 192 8 16 32 2>/dev/null >> $out/io.txt
done
cat $out/io.txt
