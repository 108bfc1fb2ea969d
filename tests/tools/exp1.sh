# developer tool: same-box A/B of update schedule / grid switches (tests/tools/upd_time.py), two rounds
set -o pipefail
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  tag=${v%%:*}; envs=${v#*:}
  env $envs KB_TAG=$tag python3 tests/tools/upd_time.py 20 2>/dev/null | tail -1
done; done
