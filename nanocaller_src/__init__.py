"""`nanocaller_src` -- the package name the reference's `NanoCaller` script imports (NanoCaller:11,13; snpCaller.py:6-10,
indelCaller.py:6-11).  It is an alias of `nanocaller_amd`: every module of the hot path is importable under the reference's
module path (`nanocaller_src.snpCaller`, `.indelCaller`, `.utils`, `.generate_SNP_pileups`, `.generate_indel_pileups`,
`.generate_indel_pileups_haploid`, `.model_architect`, `.model_architect_SNP_haploid`, `.model_architect_indel`,
`.model_architect_indels_haploid`) and is the SAME module object as its `nanocaller_amd` counterpart, so `run(args)` of the
reference's script (NanoCaller:12-56) drives the MI355X path unchanged."""
import importlib
import sys

MODULES = ("utils", "snpCaller", "indelCaller", "generate_SNP_pileups", "generate_indel_pileups", "generate_indel_pileups_haploid",
           "model_architect", "model_architect_SNP_haploid", "model_architect_indel", "model_architect_indels_haploid")

for _name in MODULES:
    _mod = importlib.import_module("nanocaller_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod          # `import nanocaller_src.<module>` finds it here
    globals()[_name] = _mod                              # `from nanocaller_src import <module>`
del _name, _mod
