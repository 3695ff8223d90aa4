"""Driver for the trainer's schedule state machines, shared by tools/make_golden.py (runs it on the REFERENCE class) and
tests/test_joint_loop.py (runs it on ours and compares with tests/golden/schedule.npz)."""
import numpy as np


def _schedule_trace(trainer_cls, make_self, n_steps):
    """Drive update_iter_type / update_explo_type the way train() does (lb_online_trainer_v7.py:493-540) with scripted buffer
    growth; returns per step (iter_type is 'vid-bias', explo_type_rand is 'explo', explo_type_vid is 'explo', do_vid, do_rand)."""
    t = make_self()
    rows = []
    for step in range(n_steps):
        t.step = step
        trainer_cls.update_iter_type(t)
        trainer_cls.update_explo_type(t)
        do_vid = step > t.init_rand_steps and step % t.video_explo_freq == 0 and t.explo_type_vid == 'explo'
        do_rand = step > t.init_rand_steps and step % t.rand_explo_freq == 0 and t.explo_type_rand == 'explo'
        if do_vid:
            t.envBuf_vid.n = min(t.envBuf_vid.n + 8, 600)
        if do_rand:
            t.envBuf_rand.n = min(t.envBuf_rand.n + 16, 1200)
        if t.iter_type == 'rand-bias':
            t.rand_iter_cnt += 1
        else:
            t.vid_iter_cnt += 1
        rows.append((t.iter_type == 'vid-bias', t.explo_type_rand == 'explo', t.explo_type_vid == 'explo', do_vid, do_rand))
    return np.array(rows, dtype=np.uint8)


class _Len:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def schedule_stub(cfg):
    """Attribute bag holding exactly the state the two schedule methods read (init_helpers, lb_online_trainer_v7.py:266-300)."""
    from types import SimpleNamespace
    return SimpleNamespace(step=0, iter_type='rand-bias', rand_iter_cnt=0, vid_iter_cnt=0, explo_type_rand='explo',
                           explo_type_vid='explo', cnt_no_exp_rand=0, cnt_exp_rand=0, cnt_no_exp_vid=0, cnt_exp_vid=0,
                           envBuf_rand=_Len(cfg['n_rand0']), envBuf_vid=_Len(0), **{k: v for k, v in cfg.items() if k != 'n_rand0'})


SCHEDULE_CFGS = {
    # values of config/libero/lb_tk8_65to72.py:84-113 (init_rand_steps shortened in the 2nd/3rd so the cycles are exercised early)
    "released": dict(init_rand_steps=10000, rand_cycle_steps=100, vid_cycle_steps=400, video_explo_freq=200, rand_explo_freq=500,
                     enable_noExp=True, noExp_start_buf_len_rand=500, noExp_start_buf_len_vid=500, Exp_noExp_rand=(1000, 1000),
                     Exp_noExp_vid=(1000, 1000), n_rand0=400),
    "short": dict(init_rand_steps=200, rand_cycle_steps=100, vid_cycle_steps=400, video_explo_freq=50, rand_explo_freq=70,
                  enable_noExp=True, noExp_start_buf_len_rand=420, noExp_start_buf_len_vid=40, Exp_noExp_rand=(300, 200),
                  Exp_noExp_vid=(250, 150), n_rand0=400),
    "vid_only": dict(init_rand_steps=0, rand_cycle_steps=0, vid_cycle_steps=10, video_explo_freq=20, rand_explo_freq=30,
                     enable_noExp=False, noExp_start_buf_len_rand=None, noExp_start_buf_len_vid=None, Exp_noExp_rand=None,
                     Exp_noExp_vid=None, n_rand0=8),
}
