"""Board telemetry beside a measurement (bench.py's roofline.power_w / power_cap_w / sclk_mhz; VERDICT r5 next #1c).

MI355X clocks to its socket power budget: what a kernel sustains is the product of its schedule AND the clock the firmware grants under that
kernel's switching activity.  A sampler thread reads the firmware's metrics table through AMD SMI (the `amdsmi` Python binding of libamd_smi.so
that ships with ROCm) while a region runs, so that "the probe reads 1.5 PFLOP/s, not 2.5" is explained by a measured clock and a measured power
against the board's cap instead of by inference.  Measurement aid only: nothing on the model path imports this module, and every failure
(no library, no permission, unknown field) degrades to `{'available': False, 'reason': ...}` - it never costs a bench line.
"""
from __future__ import annotations

import threading
import time
from typing import Dict, List, Optional

_NA = (None, 'N/A', 0xFFFF, 0xFFFFFFFF, 65535)


def _num(v) -> Optional[float]:
    if v in _NA or isinstance(v, str):
        return None
    try:
        return float(v)
    except (TypeError, ValueError):
        return None


class BoardSampler:
    """with BoardSampler(device_index) as s: <region>; s.summary() -> averages over the region.

    Per sample: socket power (W), the gfx clock of every XCD (MHz), hotspot temperature, and the firmware's accumulated throttle-residency
    counters (power / thermal), whose growth over the region says how much of it ran power-limited."""

    def __init__(self, device_index: int = 0, period_s: float = 0.02):
        self.period = period_s
        self.samples: List[Dict[str, float]] = []
        self.err: Optional[str] = None
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self._h = None
        self._smi = None
        self.cap_w: Optional[float] = None
        self.max_clk: Optional[float] = None
        try:
            import amdsmi
            self._smi = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            if not hs:
                raise RuntimeError('no AMD SMI processor handles')
            self._h = hs[min(device_index, len(hs) - 1)]
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(self._h)
                c = _num(cap.get('power_cap'))
                if c:
                    self.cap_w = c / 1e6 if c > 1e5 else c          # microwatts in the library's own unit, watts in some versions
            except Exception:
                pass
            if self.cap_w is None:
                try:
                    c = _num(amdsmi.amdsmi_get_power_info(self._h).get('power_limit'))
                    if c:
                        self.cap_w = c / 1e6 if c > 1e5 else c
                except Exception:
                    pass
            try:
                ci = amdsmi.amdsmi_get_clock_info(self._h, amdsmi.AmdSmiClkType.GFX)
                self.max_clk = _num(ci.get('max_clk'))
            except Exception:
                pass
        except Exception as e:                                       # noqa: BLE001 - a measurement aid must never raise into the bench
            self.err = f'{type(e).__name__}: {e}'[:200]

    # ------------------------------------------------------------------------------------------------------------------
    def _read(self) -> Optional[Dict[str, float]]:
        smi, h = self._smi, self._h
        out: Dict[str, float] = {}
        try:
            m = smi.amdsmi_get_gpu_metrics_info(h)
            p = _num(m.get('current_socket_power')) or _num(m.get('average_socket_power'))
            if p is not None:
                out['power_w'] = p
            clks = [c for c in (_num(v) for v in (m.get('current_gfxclks') or [])) if c]
            if clks:
                out['sclk_mhz'] = sum(clks) / len(clks)
                out['sclk_min_xcd'] = min(clks)
            elif _num(m.get('current_gfxclk')):
                out['sclk_mhz'] = _num(m.get('current_gfxclk'))
            for k in ('temperature_hotspot', 'ppt_residency_acc', 'prochot_residency_acc', 'socket_thm_residency_acc', 'vr_thm_residency_acc',
                      'hbm_thm_residency_acc', 'accumulation_counter', 'average_gfx_activity'):
                v = _num(m.get(k))
                if v is not None:
                    out[k] = v
        except Exception as e:                                       # noqa: BLE001
            if self.err is None:
                self.err = f'gpu_metrics: {type(e).__name__}: {e}'[:200]
        if 'power_w' not in out:
            try:
                pi = smi.amdsmi_get_power_info(h)
                p = _num(pi.get('current_socket_power')) or _num(pi.get('socket_power')) or _num(pi.get('average_socket_power'))
                if p is not None:
                    out['power_w'] = p
            except Exception:
                pass
        if 'sclk_mhz' not in out:
            try:
                c = _num(smi.amdsmi_get_clock_info(h, smi.AmdSmiClkType.GFX).get('clk'))
                if c:
                    out['sclk_mhz'] = c
            except Exception:
                pass
        return out or None

    def _loop(self):
        while not self._stop.is_set():
            s = self._read()
            if s:
                s['t'] = time.perf_counter()
                self.samples.append(s)
            self._stop.wait(self.period)

    def start(self):
        if self._h is None:
            return self
        self.samples = []
        self._stop.clear()
        self._thread = threading.Thread(target=self._loop, name='cvar-board-sampler', daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=2.0)
            self._thread = None
        return self

    __enter__ = start

    def __exit__(self, *exc):
        self.stop()
        return False

    # ------------------------------------------------------------------------------------------------------------------
    def summary(self, skip_first_s: float = 0.0) -> Dict[str, object]:
        """averages over the samples of the region (optionally without its first `skip_first_s` seconds: the clock ramps for ~0.1 s)"""
        if self._h is None:
            return {'available': False, 'reason': self.err or 'AMD SMI unavailable'}
        ss = self.samples
        if ss and skip_first_s > 0:
            t0 = ss[0]['t'] + skip_first_s
            ss = [s for s in ss if s['t'] >= t0] or ss
        if not ss:
            return {'available': False, 'reason': self.err or 'no sample inside the region'}

        def col(k):
            return [s[k] for s in ss if k in s]

        def avg(k, nd=1):
            v = col(k)
            return round(sum(v) / len(v), nd) if v else None

        out: Dict[str, object] = {'available': True, 'samples': len(ss), 'seconds': round(ss[-1]['t'] - ss[0]['t'], 3),
                                  'power_w': avg('power_w'), 'power_w_max': round(max(col('power_w')), 1) if col('power_w') else None,
                                  'power_cap_w': None if self.cap_w is None else round(self.cap_w, 1),
                                  'sclk_mhz': avg('sclk_mhz', 0), 'sclk_mhz_min': round(min(col('sclk_mhz')), 0) if col('sclk_mhz') else None,
                                  'sclk_mhz_max': round(max(col('sclk_mhz')), 0) if col('sclk_mhz') else None, 'sclk_limit_mhz': self.max_clk,
                                  'hotspot_c': avg('temperature_hotspot', 0)}
        if out['power_w'] is not None and self.cap_w:
            out['power_frac_of_cap'] = round(float(out['power_w']) / self.cap_w, 3)
        # throttle residency: the firmware accumulates, per 1 ms tick of `accumulation_counter`, whether each limiter was active
        acc = col('accumulation_counter')
        if len(acc) >= 2 and acc[-1] > acc[0]:
            for k, name in (('ppt_residency_acc', 'power_limited_frac'), ('socket_thm_residency_acc', 'thermal_limited_frac'), ('prochot_residency_acc', 'prochot_frac')):
                v = col(k)
                if len(v) == len(acc):
                    out[name] = round((v[-1] - v[0]) / (acc[-1] - acc[0]), 3)
        if self.err:
            out['note'] = self.err
        return out
