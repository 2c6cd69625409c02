"""Mirror of the reference's ``detr_od/models/utils/ops`` package layout (functions/, modules/)."""
from .functions import MSDeformAttnFunction, MSDeformAttnFusedFunction  # noqa: F401
from .modules import MSDeformAttn  # noqa: F401
