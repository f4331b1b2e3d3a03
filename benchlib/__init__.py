"""Pieces of bench.py (the repo-root benchmark driver): one module per BASELINE configuration, the self-launcher of the
rank processes, and what they share (peaks, the committed PMC summaries, small helpers). bench.py parses the command
line and dispatches."""
