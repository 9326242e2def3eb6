"""Model registry (reference: models/__init__.py:27-69): `--model NAME` -> class NAMEModel in NAME_model.py."""
import importlib

from .base_model import BaseModel


def find_model_using_name(model_name):
    model_filename = 'neuroclear_amd.models.' + model_name + '_model'
    try:
        modellib = importlib.import_module(model_filename)
    except ImportError:
        raise NotImplementedError('model [%s] is not part of the MI355X hot path (available: test, '
                                  'axial_to_lateral_gan_apollo, axial_to_lateral_gan_athena, '
                                  'axial_to_lateral_gan_dryops)' % model_name)
    target = model_name.replace('_', '') + 'model'
    for name, cls in modellib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseModel):
            return cls
    raise NotImplementedError('In %s.py, there should be a subclass of BaseModel named %s' % (model_filename, target))


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    instance = find_model_using_name(opt.model)(opt)
    print('model [%s] was created' % type(instance).__name__)
    return instance
