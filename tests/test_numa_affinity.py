"""Optional rank -> CPU affinity by the GPU's NUMA node (mri_inr_amd/launch.py, `bench.py --numa-pin`): a fake sysfs tree and a
fake `rocm-smi --showbus --json`.  No multi-GPU node was available to this build: the logic is tested, no effect is claimed."""
import json
import os

from mri_inr_amd import launch


def _fake_sysfs(root, gpus, nodes):
    for busid, node in gpus.items():
        d = root / "bus" / "pci" / "devices" / busid
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    for node, cpulist in nodes.items():
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpulist + "\n")
    return str(root)


def test_cpulist_round_trip():
    assert launch.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert launch.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11"
    assert launch.parse_cpulist("") == [] and launch.format_cpulist([]) == ""


def test_busids_from_smi_json_and_garbage():
    good = json.dumps({"card1": {"PCI Bus": "0000:15:00.0"}, "card0": {"PCI Bus": "0000:05:00.0"}, "system": {"Driver version": "6"}})
    assert launch.gpu_busids_from_smi(lambda: good) == ["0000:05:00.0", "0000:15:00.0"]
    assert launch.gpu_busids_from_smi(lambda: "not json") == []
    assert launch.gpu_busids_from_smi(lambda: json.dumps({"card0": {"PCI Bus": "N/A"}})) == []

    def missing():
        raise FileNotFoundError("rocm-smi")
    assert launch.gpu_busids_from_smi(missing) == []


def test_rank_affinity_from_a_fake_two_socket_tree(tmp_path):
    gpus = {"0000:05:00.0": 0, "0000:15:00.0": 0, "0000:85:00.0": 1, "0000:95:00.0": -1}
    sysfs = _fake_sysfs(tmp_path, gpus, {0: "0-47,96-143", 1: "48-95,144-191"})
    ids = list(gpus)
    a = launch.rank_affinity(0, busids=ids, sysfs=sysfs, allowed=range(192), env={})
    assert a == {"pci_bus_id": "0000:05:00.0", "numa_node": 0, "cpus": "0-47,96-143"}
    b = launch.rank_affinity(2, busids=ids, sysfs=sysfs, allowed=range(192), env={})
    assert b["numa_node"] == 1 and b["cpus"] == "48-95,144-191"
    # the mask the process already has (a cgroup share, taskset) is respected
    c = launch.rank_affinity(2, busids=ids, sysfs=sysfs, allowed=range(40, 64), env={})
    assert c["cpus"] == "48-63"
    # unknown node, unknown device, no CPU left: nothing is pinned
    assert launch.rank_affinity(3, busids=ids, sysfs=sysfs, allowed=range(192), env={}) is None
    assert launch.rank_affinity(7, busids=ids, sysfs=sysfs, allowed=range(192), env={}) is None
    assert launch.rank_affinity(2, busids=ids, sysfs=sysfs, allowed=range(0, 8), env={}) is None
    # the bus ids may come from the environment; HIP_VISIBLE_DEVICES renumbers the devices
    env = {"MSIREN_RANK_PCI_BUSIDS": ",".join(ids), "HIP_VISIBLE_DEVICES": "2,0"}
    d = launch.rank_affinity(0, sysfs=sysfs, allowed=range(192), env=env)
    assert d["pci_bus_id"] == "0000:85:00.0" and d["numa_node"] == 1


def test_pin_rank_applies_the_mask_and_reports_it(tmp_path):
    sysfs = _fake_sysfs(tmp_path, {"0000:05:00.0": 1}, {1: "2-5"})
    seen = []
    info = launch.pin_rank(0, busids=["0000:05:00.0"], sysfs=sysfs, allowed=range(16), env={}, setter=seen.append)
    assert info == {"pci_bus_id": "0000:05:00.0", "numa_node": 1, "cpus": "2-5"} and seen == [{2, 3, 4, 5}]

    def refuse(_):
        raise OSError("EPERM")
    assert launch.pin_rank(0, busids=["0000:05:00.0"], sysfs=sysfs, allowed=range(16), env={}, setter=refuse) is None
    # on this (single-socket, GPU-less) container: whatever the real tree says, the call must not raise
    launch.rank_affinity(0, busids=["0000:00:00.0"])
    if hasattr(os, "sched_getaffinity"):
        assert os.sched_getaffinity(0)   # untouched
