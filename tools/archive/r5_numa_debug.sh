#!/bin/bash
# where the reading threads run: the box's NUMA layout, the device's node, and the CPUs of the load's and of a one-turn stream's pread threads
mkdir -p gpurun_out/r5
{
  lscpu | grep -E "NUMA|Socket|Model name|^CPU\(s\)"
  for d in /sys/bus/pci/devices/*; do if [ -e $d/numa_node ] && grep -qi 0x1002 $d/vendor 2>/dev/null && grep -q "^0x03\|^0x12" $d/class 2>/dev/null; then echo "$d numa_node=$(cat $d/numa_node)"; fi; done
  cat /sys/devices/system/node/node*/cpulist
  nproc; taskset -p $$
} > gpurun_out/r5/numa_box.txt 2>&1
DFDB_AB_REPEATS=2 DFDB_STREAM_DEBUG=1 DFDB_STREAM_DEBUG_CPUS=1 timeout 300 python tools/r5_reader_ab.py > gpurun_out/r5/numa_ab.out 2> gpurun_out/r5/numa_ab.err
cat gpurun_out/r5/numa_ab.out
