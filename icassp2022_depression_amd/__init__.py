"""Import alias: the product package lives in the directory `icassp2022-depression_amd/` (the
name the project layout prescribes; a hyphen is not importable), so this importable package
simply extends its search path to that directory.  `import icassp2022_depression_amd.audio_gru_whole`
therefore loads `icassp2022-depression_amd/audio_gru_whole.py`."""
import os as _os

_impl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'icassp2022-depression_amd')
if not _os.path.isdir(_impl):
    raise ImportError(f'product package directory missing: {_impl}')
__path__.append(_impl)
