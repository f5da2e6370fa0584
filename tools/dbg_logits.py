import json, os, sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import golden_cases as GC
import ctranslate2_amd as ct2
from ctranslate2_amd import model_builder as mb
g=json.load(open('tests/golden/model_golden.json'))
key='small'
path='/tmp/ct2x_test_small'
if not os.path.exists(path+'/model.bin'): mb.build_transformer(path, **GC.SYNTHETIC[key]['model'])
t=ct2.Translator(path, compute_type='int8')
tf=g[key]['teacher_forced']
lg=t.step_logits(g[key]['sources'], np.array(tf['tgt_in'],np.int32))
cols=np.array(tf['cols']); ref=np.array(tf['values'],np.float32)
err=np.abs(lg[...,cols]-ref)
print(os.environ.get('CT2X_GEMM_TILE'), 'max err', err.max(), 'per-batch max', err.max(axis=(0,2)))
