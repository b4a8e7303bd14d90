"""Drop-in for ``model/net.py`` of the reference: ``from model.net import Unit2D, conv_init, import_class``."""
from stgcn_amd.modules import Unit2D, conv_init, import_class  # noqa: F401
