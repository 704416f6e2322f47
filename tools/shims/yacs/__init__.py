"""Minimal stand-in for the `yacs` package (absent from the MI355X image, no network): only what the reference's
config/stereo_human_config.py uses -- see config.py.  Put on sys.path by tools/launch_stage2.py ONLY when the real package is missing."""
