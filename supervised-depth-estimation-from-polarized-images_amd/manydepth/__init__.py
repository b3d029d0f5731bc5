"""manydepth façade: the reference's module paths, class names, constructor signatures and
state_dict keys, executed by the MI355X-native ``polardepth`` engine (hand-written HIP kernels).

Only the supervised single-frame path (``--depth_supervision_only``) that the hot path covers is
implemented; the self-supervised / multi-frame branches of the reference raise NotImplementedError.
"""
