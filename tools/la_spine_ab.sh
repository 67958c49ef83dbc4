# the panel launches under the look-ahead at N = 16384 (one evaluation at a time): spine
# workgroups (GPX_PANEL_NSPINE) x workers (GPX_PANEL_WG); tools/seq_time.py 16384 6
set -e
for cfg in "-1 -1" "3 -1" "5 -1" "-1 24" "-1 48" "3 24" "3 48" "5 48" "-1 -1"; do
  set -- $cfg
  echo "== GPX_PANEL_NSPINE=$1 GPX_PANEL_WG=$2"
  env_args=""
  [ "$1" != "-1" ] && export GPX_PANEL_NSPINE=$1 || unset GPX_PANEL_NSPINE
  [ "$2" != "-1" ] && export GPX_PANEL_WG=$2 || unset GPX_PANEL_WG
  timeout -k 10 200 python tools/seq_time.py 16384 6
done
