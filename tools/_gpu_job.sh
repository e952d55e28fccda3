python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3
bash tools/final_profiles.sh 2>&1 | tail -4 | cut -c1-600
