"""`model.*` import paths of the reference Vidi-7B CLI (Vidi_7B/inference.py:8-12), resolved to the MI355X
implementation: put vidi_amd/compat_7b on PYTHONPATH and the script runs unchanged."""
