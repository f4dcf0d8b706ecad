"""Drop-in modules with the reference's import names (`lg_parser`, `utility1.*`, plus the `utility` / `world` aliases
the north star mentions).  `python -m spex_amd.dropin <reference main script> [its flags]` runs an unmodified
reference driver (e.g. LightGCN_SPEX/code/main_rec.py) on top of them."""
import os

PATH = os.path.dirname(os.path.abspath(__file__))
