# Chunk sizes of the host-buffer calls (RCX_HOST_ENC_CHUNK / RCX_HOST_DEC_CHUNK, blocks per chunk) and work streams against
# the defaults: the uniform GiB through tools/host_rate.py, one line per setting.  Run on the GPU box.
set -o pipefail
run() {
  env "$@" timeout -k 10 120 python tools/host_rate.py --workloads uniform 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$*', '| reused:', d.get('host_encode_MBps'),d.get('host_decode_MBps'),'| fresh:',d.get('host_encode_MBps_fresh_destination'),d.get('host_decode_MBps_fresh_destination'))"
}
run RCX_X=default
run RCX_HOST_DEC_CHUNK=2752 RCX_HOST_ENC_CHUNK=1408
run RCX_HOST_DEC_CHUNK=3328 RCX_HOST_ENC_CHUNK=1664
run RCX_HOST_DEC_CHUNK=2048 RCX_HOST_ENC_CHUNK=1024
run RCX_HOST_DEC_CHUNK=2752 RCX_HOST_WORK_STREAMS=3
run RCX_X=default
