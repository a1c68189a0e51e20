"""The token <-> index tables of examples/timit/metrics/mapping_files/*.txt, generated instead of shipped:
the three phone sets are the sorted columns of the Lee & Hon folding table (mapping.py _FOLD) followed by
'<' and '>', the character
sets are '_' / A-Z + a-z (+ doubled letters) + ' < >.  write_mapping_files(dir) produces phone61.txt, phone48.txt,
phone39.txt, phone2phone.txt, character.txt, character_capital_divide.txt in the reference's `<token>  <index>`
form; tests/test_host_io.py pins them to the tables read from the reference's own files
(tests/golden/labels_v1.json)."""
import os
import string

from examples.timit.metrics.mapping import _FOLD

_DOUBLED = 'abcdefgiklmnoprstuz'          # letters that also exist as a doubled token (character_capital_divide)


def phone_tables():
    p61 = sorted(_FOLD)
    p48 = sorted(set(v[0] for v in _FOLD.values() if v[0] is not None))
    p39 = sorted(set(v[1] for v in _FOLD.values() if v[1] != ''))
    return dict(phone61=p61, phone48=p48, phone39=p39)


def character_tables():
    chars = ['_'] + list(string.ascii_lowercase) + ["'", '<', '>']
    cap = list(string.ascii_uppercase)
    for c in string.ascii_lowercase:
        cap.append(c)
        if c in _DOUBLED:
            cap.append(c + c)
    cap += ["'", '<', '>']
    return dict(character=chars, character_capital_divide=cap)


def write_mapping_files(map_dir):
    os.makedirs(map_dir, exist_ok=True)
    tables = dict(phone_tables())
    tables.update(character_tables())
    for name in ('phone61', 'phone48', 'phone39'):
        tables[name] = tables[name] + ['<', '>']          # <SOS> / <EOS> of the attention models close every phone file
    for name, toks in tables.items():
        with open(os.path.join(map_dir, name + '.txt'), 'w') as f:
            for i, t in enumerate(toks):
                f.write('%s  %d\n' % (t, i))
    with open(os.path.join(map_dir, 'phone2phone.txt'), 'w') as f:
        for p61 in sorted(_FOLD):
            p48, p39 = _FOLD[p61]
            f.write('%s  %s  %s\n' % (p61, p48 if p48 is not None else 'nan', p39 if p39 != '' else 'nan'))
    return map_dir
