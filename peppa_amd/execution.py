"""`pig.execution.default_config` (pig/execution.py:4-42) and the seven published conditions
(`:44-82`).  `mi355x` is an optional extra block the reference ignores."""
from copy import deepcopy
import yaml

default_config = {
    'margin': 0.2,
    'data': {'num_workers': 12, 'extract': False, 'prepare': False, 'iterable': False,
             'target_size': [180, 100], 'audio_sample_rate': 44100,
             'train': {'force_cache': False, 'batch_size': 8, 'jitter': True, 'jitter_sd': 0.5,
                       'duration': 2.3, 'shuffle': True},
             'val': {'force_cache': False, 'batch_size': 8, 'jitter': False, 'duration': 2.3},
             'test': {'force_cache': False, 'batch_size': 8, 'jitter': False, 'duration': 2.3}},
    'video': {'pretrained': True, 'project': True, 'version': 'r2plus1d_18', 'pooling': 'attention'},
    'audio': {'path': 'data/in/wav2vec/wav2vec_small.pt', 'pretrained': True,
              'freeze_feature_extractor': False, 'freeze_encoder_layers': None,
              'pooling': 'attention', 'full': True},
    'training': {'trainer_args': {'gpus': 1, 'auto_select_gpus': False,
                                  'accumulate_grad_batches': 8, 'precision': 16}},
    'optimizer': {'lr': 0.0001, 'warmup': 0.1, 'schedule': 'warmup_linear', 't_total': 15000},
}


def conditions(base=default_config):
    config = {'base': base}
    c = deepcopy(base)
    c['audio']['freeze_feature_extractor'] = True
    c['audio']['freeze_encoder_layers'] = 12
    config['freeze_wav2vec'] = c
    c = deepcopy(base)
    c['data']['train']['jitter'] = False
    c['data']['train']['jitter_sd'] = None
    config['jitter'] = c
    c = deepcopy(base)
    c['audio']['pretrained'] = False
    config['pretraining_v'] = c
    c = deepcopy(base)
    c['video']['pretrained'] = False
    config['pretraining_a'] = c
    c = deepcopy(base)
    c['video']['pretrained'] = False
    c['audio']['pretrained'] = False
    config['pretraining_none'] = c
    c = deepcopy(base)
    c['video']['static'] = True
    del c['video']['version']
    config['static'] = c
    return config


def dump_conditions():
    for name, hparams in conditions().items():
        yaml.dump(hparams, open(f"hparams_{name}.yaml", "w"))
