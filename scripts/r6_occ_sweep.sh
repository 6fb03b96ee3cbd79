cd /root/repo
BASE_N=50000 TUNINGS=";occ_ahead_x10=30;occ_ahead_x10=50;plan_split=0;occ_stage_ahead=0;commit_par=0;commit_par=0,occ_ahead_x10=40;select_shortcut=0" python scripts/occ_at_scale.py 6144 2>&1 | grep -v "amdgpu.ids\|rounds ended\|   groups"
