"""Config contract of ``configs/main_cfg.py`` without detectron2 / omegaconf.

The reference hands each module an omegaconf node and only ever uses attribute
access and ``.get`` on it (model files).  ``Cfg`` gives
the same two behaviours over plain dicts, so a detectron2 LazyConfig's
``model.cfg`` converted with ``OmegaConf.to_container`` (or any mapping) works.
"""


class Cfg(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v

    def __setattr__(self, k, v):
        self[k] = v


def to_cfg(cfg):
    if isinstance(cfg, Cfg):
        return cfg
    if isinstance(cfg, dict):
        return Cfg(cfg)
    # omegaconf / LazyConfig nodes: mapping-like with .items()
    try:
        return Cfg({k: (dict(v) if hasattr(v, "items") else v) for k, v in cfg.items()})
    except Exception as e:  # pragma: no cover
        raise TypeError(f"unsupported cfg type {type(cfg)}") from e


def default_model_cfg(**over):
    """The ``model.cfg`` of configs/main_cfg.py:128-181 (pretrained/vit_ckpt off: no checkpoint here)."""
    cfg = dict(
        coarse_npoint=196,
        fine_npoint=2048,
        feature_extraction=dict(vit_type="vit_base_patch14_reg4_dinov2", up_type="linear", embed_dim=768, out_dim=256,
                                use_pyramid_feat=True, pretrained=False, vit_ckpt=None, freeze_vit=False, img_size=224),
        geo_embedding=dict(sigma_d=0.2, sigma_a=15, angle_k=3, reduction_a="max", hidden_dim=256),
        coarse_point_matching=dict(nblock=3, input_dim=256, hidden_dim=256, out_dim=256, temp=0.1, sim_type="cosine",
                                   normalize_feat=True, loss_predator_thres=0.15, loss_dis_thres=0.3,
                                   nproposal1=6000, nproposal2=300),
        fine_point_matching=dict(nblock=3, input_dim=256, hidden_dim=256, out_dim=256, pe_radius1=0.1, pe_radius2=0.2,
                                 focusing_factor=3, temp=0.1, sim_type="cosine", normalize_feat=True,
                                 loss_predator_thres=0.15, loss_dis_thres=0.3, use_lrf=True, use_xyz=True,
                                 nsample1=64, nsample2=256),
    )
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    return Cfg(cfg)
