set -u
SECONDS=0
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -25
echo "all $SECONDS s"
