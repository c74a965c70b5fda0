#!/usr/bin/env python3
"""Build-container script: feeds result pickles written by THIS repo's main.py to the reference's own consumers
(utils/experiment_utils.py: get_best_hp :186-247, combine_runs :250-288, get_returns :291-334;
main_concurrent.py: combine_data_dictionaries :107-154) and stores what they return as a fixture.
tests/test_pickle_consumers.py re-creates the pickles and recomputes the same quantities WITHOUT the reference.

    python tests/golden/make_pickle_golden.py        (needs /root/reference; never runs on the GPU box)
"""
import copy
import json
import os
import pickle
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("RLCONTROL_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    from pickle_fixture_common import write_pickles
    tmp = tempfile.mkdtemp(prefix="pkl_golden_")
    try:
        paths = write_pickles(tmp)
        sys.path.insert(0, REF)
        import utils.experiment_utils as eu                  # the reference (numpy only)
        import main_concurrent as mc                         # the reference (click, glob, pickle)
        datas = [pickle.load(open(p, "rb")) for p in paths]
        out = {"index_ranges": [os.path.basename(p) for p in paths]}
        # 1. one pickle on its own
        out["best_hp_eval_first"] = [[int(i), float(v)] for i, v in eu.get_best_hp(datas[0], "eval")]
        # 2. combine_runs(data1, data2): NOTE it tests `hp_setting not in data2.keys()` (top-level keys), so with the
        #    reference's own schema it raises KeyError for every real pickle -- recorded as is
        d1 = copy.deepcopy(datas[0])
        try:
            eu.combine_runs(d1, copy.deepcopy(datas[1]))
            out["combine_runs"] = {"raised": None,
                                   "runs_per_setting": {str(k): len(v["runs"]) for k, v in d1["experiment_data"].items()}}
        except KeyError as e:
            out["combine_runs"] = {"raised": "KeyError", "message": str(e)}
        # 3. combine_data_dictionaries over the directory of both pickles
        pdir = os.path.dirname(paths[0])
        comb = mc.combine_data_dictionaries(pdir)
        os.remove(os.path.join(pdir, "data.pkl"))
        out["combined_settings"] = sorted(int(k) for k in comb["experiment_data"])
        out["combined_runs_per_setting"] = {str(k): len(v["runs"]) for k, v in comb["experiment_data"].items()}
        out["combined_seeds"] = {str(k): sorted(int(r["random_seed"]) for r in v["runs"])
                                 for k, v in comb["experiment_data"].items()}
        out["best_hp_eval"] = [[int(i), float(v)] for i, v in eu.get_best_hp(comb, "eval")]
        out["best_hp_eval_after_-2"] = [[int(i), float(v)] for i, v in eu.get_best_hp(comb, "eval", after=-2)]
        out["best_hp_train"] = [[int(i), float(v)] for i, v in eu.get_best_hp(comb, "train")]
        for ind in (0, 3):
            ev = eu.get_returns(comb, "eval", ind)
            tr = eu.get_returns(comb, "train", ind)
            # runs sorted by seed: glob order (hence run order) is file-system dependent
            order = np.argsort([r["random_seed"] for r in comb["experiment_data"][ind]["runs"]])
            out["returns_eval_%d" % ind] = {"shape": list(ev.shape), "values": ev[order].tolist()}
            out["returns_train_%d" % ind] = {"shape": list(tr.shape), "values": tr[order].tolist()}
            out["hyperparams_%d" % ind] = {k: v for k, v in eu.get_hyperparams(comb, ind).items()
                                           if isinstance(v, (int, float, str))}
        with open(os.path.join(HERE, "pickle_consumers.json"), "w") as f:
            json.dump(out, f, indent=1)
        print("wrote", os.path.join(HERE, "pickle_consumers.json"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
