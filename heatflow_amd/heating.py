"""Heating curve: CSV -> amplitude(t) and the Gaussian boundary profile.

Mirrors run_with_diamond.py:251-274 (CSV clean-up: sort by time, numeric coercion,
rows with NaN time/temp dropped, ValueError when a column is missing) and :343-359
(np.interp with clamped ends, offset so the curve starts at ic_temp, Gaussian in r with
the given FWHM centred on the axis).
"""
from __future__ import annotations

import csv

import numpy as np


class HeatingCurve:
    def __init__(self, csv_path, ic_temp, fwhm, column="temp"):
        """``column`` = CSV column driving the curve: "temp" (p-side, the reference's only use) or
        "oside" (extension: the o-side face in a two-sided run)."""
        times, temps = [], []
        with open(csv_path, newline="") as f:
            rd = csv.DictReader(f)
            cols = rd.fieldnames or []
            if "temp" not in cols:
                raise ValueError(f"Heating CSV file {csv_path} must contain a 'temp' column")
            if column not in cols:
                raise ValueError(f"Heating CSV file {csv_path} must contain a '{column}' column")
            if "time" not in cols:
                raise ValueError(f"Heating CSV file {csv_path} must contain a 'time' column")
            for row in rd:
                try:
                    t, T = float(row["time"]), float(row[column])
                except (TypeError, ValueError):
                    continue
                if np.isnan(t) or np.isnan(T):
                    continue
                times.append(t)
                temps.append(T)
        if not times:
            raise ValueError(f"Heating CSV file {csv_path} holds no numeric rows")
        order = np.argsort(np.array(times), kind="stable")
        self.time = np.array(times)[order]
        self.temp = np.array(temps)[order]
        self.ic_temp = float(ic_temp)
        self.fwhm = float(fwhm)
        self.offset = self.temp[0] - self.ic_temp
        self.coeff = -4.0 * np.log(2.0) / self.fwhm ** 2
        # normalised curve the reference keeps for plotting (run_with_diamond.py:274)
        self.temp_normed = (self.temp - self.temp[0]) / (self.temp.max() - self.temp.min())

    def amplitude(self, t):
        """heating_offset(t) (run_with_diamond.py:350-351)."""
        return float(np.interp(t, self.time, self.temp, left=self.temp[0], right=self.temp[-1])) - self.offset

    def gaussian(self, x, y, t):
        """Boundary value at (z=x, r=y): (amp - ic) exp(coeff r^2) + ic (:357-359)."""
        amp = self.amplitude(t)
        return (amp - self.ic_temp) * np.exp(self.coeff * (np.asarray(y) - 0.0) ** 2) + self.ic_temp
