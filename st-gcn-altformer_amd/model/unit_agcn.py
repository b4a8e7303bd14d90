"""Drop-in for ``model/unit_agcn.py`` of the reference: ``from model.unit_agcn import unit_agcn``.

The reference file also defines three module-level init helpers, and one caller imports one of them from here
(``from model.unit_agcn import unit_agcn, conv_init``, model/ST_TR/ST_TR_new.py:8 — the kaiming fan_out form of
model/unit_agcn.py:12-14, NOT model/net.py's he-normal ``conv_init``), so they keep their names here.
(``model`` deliberately has no ``__init__.py``: the reference's ``model`` is a namespace package as well, so with this
directory first on ``sys.path`` ``model.net`` / ``model.unit_agcn`` resolve here and ``model.AltFormer`` etc. still
resolve to the reference's own files.)
"""
from stgcn_amd.modules import unit_agcn  # noqa: F401
from stgcn_amd.modules import _agcn_conv_init as conv_init  # noqa: F401   model/unit_agcn.py:12-14
from stgcn_amd.modules import _bn_init as bn_init  # noqa: F401            model/unit_agcn.py:17-19
from stgcn_amd.modules import _conv_branch_init as conv_branch_init  # noqa: F401   model/unit_agcn.py:22-28
