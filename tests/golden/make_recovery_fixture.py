"""Generates tests/golden/recovery_fixture.json by RUNNING the reference's own recoveries.py (the one module of the
reference that loads stand-alone: deps os + yaml) on a small synthetic layout. Run in the build container only
(/root/reference does not exist on the GPU box); the produced fixture is data, not code.

    python tests/golden/make_recovery_fixture.py
"""
import importlib.util
import json
import logging
import os
import tempfile

REF = "/root/reference/TreeDetection/recoveries.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    spec = importlib.util.spec_from_file_location("ref_recoveries", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    log = logging.getLogger("fixture")
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        os.makedirs("tiles")
        os.makedirs("out")
        metas = {
            "a": {"a_0_0_50_20_25832": {"only_forest": False, "only_urban": False},
                  "a_50_0_50_20_25832": {"only_forest": True, "only_urban": False},
                  "a_0_50_50_20_25832": {"only_forest": False, "only_urban": True}},
            "b": {"b_0_0_50_20_25832": {"only_forest": False, "only_urban": False},
                  "b_50_0_50_20_25832": {"only_forest": False, "only_urban": False}},
        }
        for stem, m in metas.items():
            with open(f"tiles/{stem}.json", "w") as f:
                json.dump(m, f)
        os.makedirs("out/a")
        os.makedirs("out/b")
        for k in metas["a"]:
            open(f"out/a/Prediction_{k}.json", "w").write("[]")
        open("out/b/Prediction_b_0_0_50_20_25832.json", "w").write("[]")     # b is incomplete
        ref.save_prediction_recovery_data("out", "tiles", "model.pth", {"img/b.tif"}, ["img/a.tif"])
        yaml_text = open("out/prediction_recovery.yaml").read()
        for model, exclude in (("model.pth", None), ("other.pth", None), ("model.pth", ["only_forest"])):
            fl, done = ref.load_prediction_recovery_data("out", "tiles", model, log, exclude)
            cases.append({"model": model, "exclude": exclude, "file_list": fl, "processed": sorted(done)})
        # with the exclude flag, a's expected count drops to 2: remove one output file and reload
        os.remove("out/a/Prediction_a_50_0_50_20_25832.json")
        fl, done = ref.load_prediction_recovery_data("out", "tiles", "model.pth", log, ["only_forest"])
        cases.append({"model": "model.pth", "exclude": ["only_forest"], "removed": "a_50_0_50_20_25832",
                      "file_list": fl, "processed": sorted(done)})
    with open(os.path.join(HERE, "recovery_fixture.json"), "w") as f:
        json.dump({"metas": metas, "yaml": yaml_text, "cases": cases}, f, indent=1)
    print("wrote recovery_fixture.json")


if __name__ == "__main__":
    main()
