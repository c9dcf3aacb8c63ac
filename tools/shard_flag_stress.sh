#!/bin/bash
# repeats the two-rank sharded visual-inertial solve with device-flag hand-overs (two ranks on ONE GPU, gloo): tools/shard_flag_stress.sh N [VAR=VALUE ...]
N=$1; shift
fail=0
for i in $(seq 1 $N); do
  port=$((20000 + RANDOM % 20000))
  env MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 VICALIB_AMD_SHARD_FLAG_SYNC=1 "$@" timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port $port tests/dist_worker.py gpu_imu > /tmp/sfs_$i.log 2>&1
  if [ $? -ne 0 ]; then fail=$((fail + 1)); grep -m1 -o "Max relative difference among violations: [0-9.e-]*" /tmp/sfs_$i.log; grep -c "ran into its bound" /tmp/sfs_$i.log; fi
done
echo "variant [$*]: $fail of $N runs failed"
