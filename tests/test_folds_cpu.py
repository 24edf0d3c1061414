"""Fold drivers against the reference's own loop statements: tests/golden/fold_bodies.npz was produced by
exec'ing the fold-loop bodies of Classification/audio_gru_whole.py:265-299, text_bilstm_whole.py:263-292 and
Regression/audio_bilstm_perm.py:215-240 (cut out by AST, tests/golden/make_golden.py::fold_bodies) on synthetic corpora, three
folds in sequence.  The package's fold_split() functions must build the same index lists and the same grown feature / label
arrays.  No GPU needed: the modules import without one and nothing here builds a model."""
import os

import numpy as np
import pytest

from conftest import ROOT

Z = np.load(os.path.join(ROOT, 'tests', 'golden', 'fold_bodies.npz'))


@pytest.mark.parametrize('key,modname,names', [
    ('audio_clf', 'audio_gru_whole', ('audio_features', 'audio_targets', 'audio_dep_idxs_tmp', 'audio_non_idxs')),
    ('text_clf', 'text_bilstm_whole', ('text_features', 'text_targets', 'text_dep_idxs_tmp', 'text_non_idxs'))])
def test_classification_fold_split_equals_reference_loop(key, modname, names):
    import importlib
    m = importlib.import_module('icassp2022_depression_amd.' + modname)
    saved = {n: getattr(m, n) for n in names}
    try:
        feats, targs = Z[key + '/feats'].copy(), Z[key + '/targs'].copy()
        setattr(m, names[0], feats); setattr(m, names[1], targs)
        setattr(m, names[2], np.where(targs == 1)[0]); setattr(m, names[3], np.where(targs == 0)[0])
        for k in range(3):
            tr, te = m.fold_split(Z[f'{key}/fold{k}'])
            assert list(tr) == Z[f'{key}/train{k}'].tolist(), k
            assert list(te) == Z[f'{key}/test{k}'].tolist(), k
        assert np.array_equal(getattr(m, names[0]), Z[key + '/feats_after'])        # the permuted rows themselves, in append order
        assert np.array_equal(getattr(m, names[1]), Z[key + '/targs_after'])
    finally:
        for n, v in saved.items():
            setattr(m, n, v)


def test_regression_fold_split_equals_reference_loop():
    from icassp2022_depression_amd import audio_bilstm_perm as m
    names = ('audio_features', 'audio_targets', 'dep_idxs', 'non_idxs', 'train_dep_idxs', 'train_non_idxs', 'test_dep_idxs', 'test_non_idxs')
    saved = {n: getattr(m, n) for n in names}
    try:
        m.audio_features = Z['audio_reg/feats'].copy(); m.audio_targets = Z['audio_reg/targs'].copy()
        m.dep_idxs = Z['audio_reg/dep_idxs']; m.non_idxs = Z['audio_reg/non_idxs']
        for fold in range(3):
            trd, trn, ted, ten = m.fold_split(fold)
            assert list(trd) == Z[f'audio_reg/train_dep{fold}'].tolist()
            assert list(trn) == Z[f'audio_reg/train_non{fold}'].tolist()             # list(set(...)) order included
            assert list(ted) == Z[f'audio_reg/test_dep{fold}'].tolist() and list(ten) == Z[f'audio_reg/test_non{fold}'].tolist()
            assert list(m.train_dep_idxs) == list(trd) and list(m.test_non_idxs) == list(ten)      # the globals train() / evaluate() read
        assert np.array_equal(m.audio_features, Z['audio_reg/feats_after'])
        assert np.allclose(m.audio_targets, Z['audio_reg/targs_after'], rtol=0, atol=0)
    finally:
        for n, v in saved.items():
            setattr(m, n, v)
