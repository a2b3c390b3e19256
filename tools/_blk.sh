for v in 0 70000 90000; do echo "block lds $v"; SBGPU_BLOCK_LDS=$v ./tools/sweep_phases.sh 0 ""; SBGPU_BLOCK_LDS=$v ./tools/sweep_phases.sh 0 ""; done
