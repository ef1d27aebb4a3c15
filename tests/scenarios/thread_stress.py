"""GPU box: several host threads use the library at once -- each makes evaluation plans, submits batches through them
and closes them (which retires and destroys their streams) while the others are launching -- with a ring of only 8
work-queue slots, so that slots are re-claimed across threads and streams all the time (capi.hip: get_queue_slot,
retire_stream).  Every batch must return what the first one returned.  Prints one JSON line.
usage: thread_stress.py [threads] [plans per thread] [batches per plan]"""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(nthreads=4, nplans=12, nbatches=6):
    from chain_scenario import joint_target
    from bayhunter_amd.synthetic import draw_models
    data = os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed')
    B, Lmax = 1500, 12
    H, VP, VS, RHO, nl = draw_models(B, (2, 12), seed=77, sorted_vs=False, Lmax=Lmax)
    packed = np.stack([H, VP, VS, RHO], axis=1)
    noise = np.tile([0.0, 0.012, 0.9, 0.01], (B, 1))
    joints = [joint_target(data) for _ in range(nthreads)]
    for j in joints:
        j.set_target_covariance([True, True], [0.0, 0.9], 1e-5)
    with joints[0].eval_plan(B, Lmax) as plan:
        plan.packed[:], plan.nlay[:], plan.noise[:] = packed, nl, noise
        plan.submit(B)
        want = tuple(a.copy() for a in plan.wait())
    errors, done = [], [0] * nthreads

    def worker(k):
        try:
            for _ in range(nplans):
                with joints[k].eval_plan(B, Lmax) as plan:
                    plan.packed[:], plan.nlay[:], plan.noise[:] = packed, nl, noise
                    for _ in range(nbatches):
                        plan.submit(B)
                        logL, mis = plan.wait()
                        if not (np.array_equal(logL, want[0], equal_nan=True) and np.array_equal(mis, want[1], equal_nan=True)):
                            raise AssertionError('thread %d: a batch differs' % k)
                        done[k] += 1
        except Exception as e:                      # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    print(json.dumps(dict(ok=not errors, errors=errors[:3], batches=sum(done), threads=nthreads,
                          slots=os.environ.get('BH_SWD_QUEUE_SLOTS'))))
    return 0 if not errors else 1


if __name__ == '__main__':
    sys.exit(main(*[int(a) for a in sys.argv[1:4]]))
