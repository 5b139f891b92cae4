cd /root/repo; timeout 2700 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -30
