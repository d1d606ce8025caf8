export PYTHONPATH=$PWD
python - <<'P'
import torch, bench
for j in (16, 0, 2):
    r = bench.reply_ingest_leg(torch, torch.device('cuda:0'), junk_every=j)
    print("junk_every", j, "call_us", round(r["call_us"], 1))
P
