"""`get_featurizer` / `get_feat_size` / `GenericFeaturizer` with the reference's names and meaning
(cufacesearch/cufacesearch/featurizer/generic_featurizer.py:5-71).  ``featurizer_type = "sbhip"`` selects the
MI355X DeepSentibank featurizer; "sbpycaffe" / "sbcmdline" resolve to it too, so existing configuration files
keep working; "dlib" selects the MI355X face-descriptor network."""


def get_featurizer(featurizer_type, global_conf, prefix=None):
    if featurizer_type in ("sbhip", "sbpycaffe", "sbcmdline"):
        from .sbhip_img_featurizer import SentiBankHIPImgFeaturizer
        if prefix:
            return SentiBankHIPImgFeaturizer(global_conf, prefix=prefix)
        return SentiBankHIPImgFeaturizer(global_conf)
    if featurizer_type in ("dlib", "dlibhip"):
        from .dlibhip_featurizer import DLibHIPFeaturizer
        if prefix:
            return DLibHIPFeaturizer(global_conf, prefix=prefix)
        return DLibHIPFeaturizer(global_conf)
    raise ValueError("[{}:error] Unknown 'featurizer' {}.".format("get_featurizer", featurizer_type))


def get_feat_size(featurizer_type):
    if featurizer_type in ("dlib", "dlibhip"):
        return 128
    if featurizer_type in ("sbhip", "sbpycaffe", "sbcmdline"):
        return 4096
    raise ValueError("[{}:error] Unknown 'featurizer' {}.".format("get_feat_size", featurizer_type))


class GenericFeaturizer(object):
    """Base class: configuration access in the style of cufacesearch.common.conf_reader.ConfReader
    (keys are looked up as ``<prefix><name>`` in the configuration dict)."""

    def __init__(self, global_conf_in, prefix=""):
        if isinstance(global_conf_in, str):
            import json
            with open(global_conf_in, "rt") as f:
                global_conf_in = json.load(f)
        self.global_conf = global_conf_in
        self.prefix = prefix
        self.pp = "GenericFeaturizer"
        self.verbose = int(self.get_param("verbose", 0) or 0)

    def set_pp(self, pp=""):
        self.pp = pp

    def get_param(self, param, default=None):
        return self.global_conf.get(self.prefix + param, default)

    def get_required_param(self, param):
        key = self.prefix + param
        if key not in self.global_conf:
            raise ValueError("[{}: error] '{}' not found in configuration".format(self.pp, key))
        return self.global_conf[key]

    def featurize(self, img, bbox=None, img_type="scikit", sha1=None):
        raise NotImplementedError("[{}:error] 'featurize' method was not overridden.".format(self.pp))
