#!/bin/bash
export TMPDIR=/tmp
bash tools/final_measure_r04.sh r04m 2 > gpurun_out/r04m_part2.log 2>&1; tail -5 gpurun_out/r04m_part2.log
echo DONE
