"""the result pickles this repo's main.py writes are accepted by the reference's OWN
offline consumers.  tests/golden/make_pickle_golden.py (build container) ran utils/experiment_utils.py
(get_best_hp :186-247, combine_runs :250-288, get_returns :291-334, get_hyperparams :337-352) and
main_concurrent.combine_data_dictionaries (:107-154) on pickles written here and stored what they returned; this test
re-creates the same pickles and recomputes those quantities from the pickle alone -- dtype / shape problems
(np.stack over runs, mean over eval episodes) or a schema drift would change them."""
import json
import os
import pickle

import numpy as np
import pytest

from pickle_fixture_common import write_pickles


@pytest.fixture(scope="module")
def fixture(golden_dir):
    with open(os.path.join(golden_dir, "pickle_consumers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def pickles(tmp_path_factory):
    paths = write_pickles(str(tmp_path_factory.mktemp("pkl")))
    return [pickle.load(open(p, "rb")) for p in paths]


def _best_hp(data, type_, after=0):
    """get_best_hp restated: mean over eval episodes, then over runs[after:], then over time; ascending argsort"""
    key = "train_episode_rewards" if type_ == "train" else "eval_episode_rewards"
    means = []
    for hp in sorted(data["experiment_data"]):
        r = np.stack([run[key] for run in data["experiment_data"][hp]["runs"]])
        if type_ == "eval":
            r = r.mean(axis=-1)
        means.append(r[after:, :].mean(axis=0).mean(axis=0))
    order = np.argsort(means)
    return [[int(i), float(means[i])] for i in order]


def _combine(datas):
    """combine_data_dictionaries restated: runs of equal settings are concatenated"""
    out = {"experiment_data": {}}
    for d in datas:
        for k, v in d["experiment_data"].items():
            if k in out["experiment_data"]:
                out["experiment_data"][k]["runs"].extend(v["runs"])
            else:
                out["experiment_data"][k] = {"agent_params": v["agent_params"], "runs": list(v["runs"])}
    return out


def _same_ranking(got, want):
    assert [v for _, v in got] == pytest.approx([v for _, v in want], rel=1e-12)
    # ties (equal means) may be ordered either way by argsort
    for (gi, gv), (wi, wv) in zip(got, want):
        assert gi == wi or any(abs(v2 - gv) < 1e-9 and i2 == gi for i2, v2 in want)


def test_single_pickle_ranks_like_the_reference(fixture, pickles):
    _same_ranking(_best_hp(pickles[0], "eval"), fixture["best_hp_eval_first"])


def test_combined_pickles_match_the_reference_consumers(fixture, pickles):
    comb = _combine(pickles)
    assert sorted(comb["experiment_data"]) == fixture["combined_settings"]
    assert {str(k): len(v["runs"]) for k, v in comb["experiment_data"].items()} == fixture["combined_runs_per_setting"]
    assert {str(k): sorted(int(r["random_seed"]) for r in v["runs"])
            for k, v in comb["experiment_data"].items()} == fixture["combined_seeds"]
    _same_ranking(_best_hp(comb, "eval"), fixture["best_hp_eval"])
    _same_ranking(_best_hp(comb, "eval", after=-2), fixture["best_hp_eval_after_-2"])
    _same_ranking(_best_hp(comb, "train"), fixture["best_hp_train"])
    for ind in (0, 3):
        runs = sorted(comb["experiment_data"][ind]["runs"], key=lambda r: r["random_seed"])
        ev = np.stack([r["eval_episode_rewards"] for r in runs])
        tr = np.expand_dims(np.stack([r["train_episode_rewards"] for r in runs]), axis=2)
        fe, ft = fixture["returns_eval_%d" % ind], fixture["returns_train_%d" % ind]
        assert list(ev.shape) == fe["shape"] and list(tr.shape) == ft["shape"]       # (runs, evals, eval episodes)
        assert np.allclose(ev, np.array(fe["values"]), rtol=1e-12) and np.allclose(tr, np.array(ft["values"]), rtol=1e-12)
        hp = comb["experiment_data"][ind]["agent_params"]
        for k, v in fixture["hyperparams_%d" % ind].items():
            assert hp[k] == v, k


def test_arrays_have_the_dtypes_the_consumers_stack(pickles):
    run = pickles[0]["experiment_data"][0]["runs"][0]
    ev = np.asarray(run["eval_episode_rewards"])
    assert ev.dtype == np.float64 and ev.shape == (4, 3)            # evaluations at 0, 400, 800, 1200 x 3 episodes
    assert np.asarray(run["train_episode_rewards"]).ndim == 1 and np.asarray(run["timesteps_at_eval"]).tolist() == [0, 400, 800, 1200]
    assert isinstance(run["random_seed"], (int, np.integer)) and isinstance(run["total_train_episodes"], (int, np.integer))


def test_reference_combine_runs_rejects_its_own_schema(fixture):
    """utils/experiment_utils.py:281 tests `hp_setting not in data2.keys()` (top-level keys), so combine_runs raises
    KeyError on every pickle main.py writes -- recorded from the reference itself, nothing for the drop-in to fix;
    main_concurrent.combine_data_dictionaries is the path that works"""
    assert fixture["combine_runs"]["raised"] == "KeyError"
