#!/bin/bash
# GPU box: the public C++ API (lyra_amd/batch_bench) on contexts that split their batches, alternating
pr() { python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    t=d.get('two_deep',{})
    print('$1 rate %d loss %d: blocking enc %.2f dec %.2f both %.2f | pipelined enc %.2f dec %.2f both %.2f M' % (d['sample_rate_hz'], d['loss_percent'], d['encode_frames_per_s']/1e6, d['decode_frames_per_s']/1e6, d['encode_decode_pipelined_frames_per_s']/1e6, t.get('encode_frames_per_s',0)/1e6, t.get('decode_frames_per_s',0)/1e6, t.get('encode_decode_two_threads_frames_per_s',0)/1e6))
"; }
for i in 1 2; do
for sb in 1 2; do
for a in "16000 0" "48000 10"; do set -- $a
LYRA_HIP_SUBBATCHES=$sb timeout 300 lyra_amd/batch_bench lyra_amd/assets 4096 $1 9200 $2 200 2>/dev/null | pr "split $sb, 4096 streams"
done
LYRA_HIP_SUBBATCHES=$sb timeout 300 lyra_amd/batch_bench lyra_amd/assets 8192 16000 9200 0 100 2>/dev/null | pr "split $sb, 8192 streams"
done; done
