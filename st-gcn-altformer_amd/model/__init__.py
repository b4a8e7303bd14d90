"""Drop-in for the two hot-path files of the reference's ``model`` package (unit_agcn.py, net.py)."""
