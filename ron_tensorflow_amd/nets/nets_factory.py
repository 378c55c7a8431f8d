"""Network factory with the interface of the reference's ``nets/nets_factory.py:34-93``."""
import functools

from . import ron_vgg_320, ssd_vgg_512

# nets/nets_factory.py:34-52, the entries of the two networks this package builds (vgg_a / vgg_16 / vgg_19 classifiers and SSD-300
# are outside SURVEY.md section 8)
networks_map = {'ron_320_vgg': ron_vgg_320.ron_net,
                'ssd_512_vgg': ssd_vgg_512.ssd_net,
                'ssd_512_vgg_caffe': ssd_vgg_512.ssd_net,
                }

arg_scopes_map = {'ron_320_vgg': ron_vgg_320.ron_arg_scope,
                  'ssd_512_vgg': ssd_vgg_512.ssd_arg_scope,
                  'ssd_512_vgg_caffe': ssd_vgg_512.ssd_arg_scope_caffe,
                  }

networks_obj = {'ron_320_vgg': ron_vgg_320.RONNet,
                'ssd_512_vgg': ssd_vgg_512.SSDNet}

# name -> variant built by the *function* entry of the reference: networks_map['ron_320_vgg'] is
# ron_vgg_320.ron_net, the full VGG-16 fc6/fc7 body (nets_factory.py:37).
_fn_variant = {'ron_320_vgg': 'full', 'ssd_512_vgg': 'ssd512'}


def get_network(name):
    """Get a network object (class) from a name (nets_factory.py:54-58)."""
    return networks_obj[name]


def get_network_fn(name, num_classes, is_training=False, **kwargs):
    """Returns network_fn(images) -> the network outputs, with `.default_image_size` (nets_factory.py:67-93).

    Extra keyword arguments: weights (dict by TF variable name; required before the first call), dtype,
    max_batch.  Raises ValueError for an unknown name exactly like the reference."""
    if name not in _fn_variant:
        raise ValueError('Name of network unknown %s' % name)
    cls = networks_obj[name]
    params = cls.default_params._replace(num_classes=num_classes)
    if name == 'ssd_512_vgg':
        net = cls(params, dtype=kwargs.pop('dtype', 'bf16'), max_batch=kwargs.pop('max_batch', 16))
    else:
        net = cls(params, variant=_fn_variant[name], dtype=kwargs.pop('dtype', 'bf16'),
                  max_batch=kwargs.pop('max_batch', 32))
    weights = kwargs.pop('weights', None)
    if weights is not None:
        net.load_weights(weights)

    @functools.wraps(net.net)
    def network_fn(images, **kw):
        return net.net(images, is_training=is_training, **kw)
    network_fn.default_image_size = params.img_shape[0]
    network_fn.network = net
    return network_fn
