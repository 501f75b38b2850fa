#!/usr/bin/env python3
"""Generate tests/golden/lengths_{charades,anet}.npz: the LENGTH statistics of the reference's own training annotations
(/root/reference/data/{charades_re0,anet_gt}/train.json) - numbers only, no text travels.

Runs only in the build container (needs /root/reference).  Re-run:  python scripts/gen_lengths.py

Per annotation (one row per training query, in file order):
  nwords    int16   words in the query.  The reference tokenises with nltk.word_tokenize(sentence.strip().lower())
                    (utils/data_gen.py:24; nltk is absent here): restated below as the Treebank rules that matter for LENGTHS
                    (punctuation split off, clitics 's n't 're 've 'll 'd 'm split, a sentence-final period split).  The
                    reference truncates to max_vlen words (data_gen.py:106 passes configs.model.max_vlen as max_pos_len),
                    so nothing is cut at the YAML settings (64 / 100) except a handful of ActivityNet captions.
  maxchars  int16   longest word of the query in characters (pad_char_seq pads a batch to its longest word,
                    utils/data_utils.py:143-155)
  charlens  uint8 [N, 24] -> not stored: the loaders only need the per-query maximum for the padded shape
  duration  float32 seconds (the annotation's own field)
  vid       int32   index of the annotation's video among the file's distinct videos (queries of one video share v_len)

v_len is NOT in the annotations: it comes from <feature_path>/feature_shapes.json (data_gen.py:175-178), which is among the
blobs this checkout lacks (.MISSING_LARGE_BLOBS / SURVEY F11).  tests/al_synth.py::make_trainset_from_lengths therefore derives
it from the duration with a stated feature rate (frames of 16 at 25 fps for ActivityNet's C3D-style extraction, 8 at 24 fps for
Charades' I3D) and clamps to max_vlen exactly as data_gen.py:178 does.
"""
import json
import os
import re

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')

_CLITIC = re.compile(r"(?i)([a-z])('s|'re|'ve|'ll|'d|'m|n't)\b")
_PUNCT = re.compile(r"([,;:@#$%&?!\"()\[\]{}<>]|\.\.\.|--|``|'')")


def tokenize(sentence):
    """the length-relevant subset of nltk's TreebankWordTokenizer as word_tokenize applies it to one lower-cased caption"""
    s = sentence.strip().lower()
    s = _PUNCT.sub(r' \1 ', s)
    s = _CLITIC.sub(r"\1 \2", s)
    toks = []
    for w in s.split():
        # word-final periods: split at the end of the sentence (Punkt + Treebank leave abbreviations inside a sentence alone)
        toks.append(w)
    if toks and len(toks[-1]) > 1 and toks[-1].endswith('.'):
        toks[-1:] = [toks[-1][:-1], '.']
    # single quotes around words
    out = []
    for w in toks:
        if len(w) > 1 and w.startswith("'") and w not in ("'s", "'re", "'ve", "'ll", "'d", "'m"):
            out += ["'", w[1:]]
        else:
            out.append(w)
    return [w for w in out if w]


def lengths(path, cap):
    data = json.load(open(path))
    vids = {}
    nwords, maxchars, dur, vid = [], [], [], []
    for rec in data:
        words = tokenize(rec[3])[:cap]
        if not words:
            words = ['.']
        nwords.append(len(words))
        maxchars.append(max(len(w) for w in words))
        dur.append(float(rec[1]))
        vid.append(vids.setdefault(rec[0], len(vids)))
    return dict(nwords=np.array(nwords, dtype=np.int16), maxchars=np.array(maxchars, dtype=np.int16),
                duration=np.array(dur, dtype=np.float32), vid=np.array(vid, dtype=np.int32))


def main():
    for task, sub, cap in (('charades', 'charades_re0', 64), ('anet', 'anet_gt', 100)):
        a = lengths(os.path.join(REF, 'data', sub, 'train.json'), cap)
        np.savez_compressed(os.path.join(OUT, 'lengths_%s.npz' % task), **a)
        nw, mc = a['nwords'], a['maxchars']
        print('%s: %d queries, %d videos; words mean %.1f max %d; longest word mean %.1f max %d; duration mean %.1f s'
              % (task, len(nw), int(a['vid'].max()) + 1, nw.mean(), nw.max(), mc.mean(), mc.max(), a['duration'].mean()))


if __name__ == '__main__':
    main()
