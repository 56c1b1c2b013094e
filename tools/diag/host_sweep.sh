# Which way across the link (RCX_HOST_IN / RCX_HOST_OUT), how many host threads, how many work streams: the uniform GiB
# through the host-buffer calls (tools/host_rate.py), one line per setting.  Run on the GPU box.
set -o pipefail
run() {
  env "$@" timeout -k 10 120 python tools/host_rate.py --workloads uniform 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$*', '| reused:', d.get('host_encode_MBps'),d.get('host_decode_MBps'),'| fresh:',d.get('host_encode_MBps_fresh_destination'),d.get('host_decode_MBps_fresh_destination'))"
}
for PF in 0 6; do
run RCX_HOST_PREFAULT=$PF RCX_HOST_OUT=staged RCX_HOST_DRAINERS=3
run RCX_HOST_PREFAULT=$PF RCX_HOST_OUT=staged RCX_HOST_DRAINERS=5
run RCX_HOST_PREFAULT=$PF RCX_HOST_OUT=direct RCX_HOST_DRAINERS=1
run RCX_HOST_PREFAULT=$PF RCX_HOST_IN=staged RCX_HOST_FEEDERS=3 RCX_HOST_OUT=staged RCX_HOST_DRAINERS=4
done
