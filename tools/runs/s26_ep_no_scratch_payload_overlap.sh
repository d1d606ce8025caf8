# round 6: (a) ep_cluster_tick_pm_kernel without its two scratch objects (the own-proposal pointer, the submission-list array): device tests + the leg (s8_ep_full.sh)
#          (b) ps_put_deliver_kernel with the batch loads ahead of the decision barrier (s25_payload_deliver_overlap.sh)
bash tools/runs/s8_ep_full.sh s26
bash tools/runs/s25_payload_deliver_overlap.sh
