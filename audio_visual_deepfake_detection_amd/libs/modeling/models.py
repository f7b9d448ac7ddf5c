"""Plugin registries of the model side -- same decorator / builder names as the reference's
libs/modeling/models.py:3-51 so `make_meta_arch(cfg['model_name'], **cfg['model'])`
(train.py:98, inference.py:63) resolves to the MI355X implementations."""

backbones, necks, generators, meta_archs = {}, {}, {}, {}


def _register(table, name):
    def deco(cls):
        table[name] = cls
        return cls
    return deco


def register_backbone(name):
    return _register(backbones, name)


def register_neck(name):
    return _register(necks, name)


def register_generator(name):
    return _register(generators, name)


def register_meta_arch(name):
    return _register(meta_archs, name)


def make_backbone(name, **kwargs):
    return backbones[name](**kwargs)


def make_neck(name, **kwargs):
    return necks[name](**kwargs)


def make_generator(name, **kwargs):
    return generators[name](**kwargs)


def make_meta_arch(name, **kwargs):
    return meta_archs[name](**kwargs)
