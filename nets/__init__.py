"""Import-path compatibility with the reference: ``from nets.pips import Pips`` (demo.py:9, chain_demo.py:9,
test_on_flt.py:6, test_on_crohd.py:6, test_on_badja.py:6, test_on_davis.py:7) resolves to pips_amd.Pips."""
