timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
