"""Registration under the reference's plugin names, for trees where mmcv / mmdet are installed.

The reference instantiates everything through mmcv registries and ``type=`` strings in its configs
(``@BBOX_ASSIGNERS.register_module()`` hungarian_assigner.py:16, o2m_assigner.py:17, ``@MATCH_COST.register_module()``
match_cost.py:8,53,146, ``@HOOKS.register_module()`` mean_teacher.py:7).  ``register_all(force=True)``
replaces those entries with the gfx950 implementations so ``configs/dino_detr`` and ``configs/detr_ssod``
run unchanged.  mmcv/mmdet are NOT installed in the build image, so this module only does something where
they exist; importing it elsewhere is harmless (returns the list of names it could not register).
"""


def register_all(force=True):
    from .matcher import BBoxL1Cost, FocalLossCost, HungarianAssigner, IoUCost, O2MAssigner
    from .mean_teacher import MeanTeacher

    done, skipped = [], []
    try:
        from mmdet.core.bbox.builder import BBOX_ASSIGNERS
        from mmdet.core.bbox.match_costs.builder import MATCH_COST
        BBOX_ASSIGNERS.register_module(name="HungarianAssigner", force=force, module=HungarianAssigner)
        BBOX_ASSIGNERS.register_module(name="O2MAssigner", force=force, module=O2MAssigner)
        for cls in (BBoxL1Cost, FocalLossCost, IoUCost):
            MATCH_COST.register_module(name=cls.__name__, force=force, module=cls)
        done += ["HungarianAssigner", "O2MAssigner", "BBoxL1Cost", "FocalLossCost", "IoUCost"]
    except ImportError:
        skipped += ["HungarianAssigner", "O2MAssigner", "BBoxL1Cost", "FocalLossCost", "IoUCost"]
    try:
        from mmdet.models.builder import LOSSES
        from .losses import TaskAlignedFocalLoss
        LOSSES.register_module(name="TaskAlignedFocalLoss", force=force, module=TaskAlignedFocalLoss)
        done.append("TaskAlignedFocalLoss")
    except ImportError:
        skipped.append("TaskAlignedFocalLoss")
    try:
        from mmcv.runner.hooks import HOOKS
        HOOKS.register_module(name="MeanTeacher", force=force, module=MeanTeacher)
        done.append("MeanTeacher")
    except ImportError:
        skipped.append("MeanTeacher")
    try:       # so that mmcv's is_module_wrapper() (runner, hooks, checkpoint code) unwraps the DDP replacement
        from mmcv.parallel import MODULE_WRAPPERS
        from .dp import FlatDDP
        MODULE_WRAPPERS.register_module(name="FlatDDP", force=force, module=FlatDDP)
        done.append("FlatDDP")
    except ImportError:
        skipped.append("FlatDDP")
    return done, skipped
