"""Host-side mirror of the reference's 13-coefficient polynomial jet model
(utils/src/JetModel.cpp:10-114, utils/include/JetModel.h).  Same method names, same units:
thrust/thrust-rate in N, N/s are standardised with (mu_T, sigma_T); throttle in percent with
(mu_u, sigma_u).  Used by the workload generator and the tick state machine; the device kernels
carry their own copy of the coefficients (csrc/vsmpc_device.hpp).
"""
from __future__ import annotations

import numpy as np


class JetModel:
    u2TCoeff = np.array([
        -4.64730485e-01, -8.13171858e+00, -6.19539230e+00, 6.61113140e-01, 1.67673231e+00,
        -4.83287064e-01, 8.77996617e+00, -1.01096376e+00, -5.86442286e-01, 5.19093322e-01,
        -4.23782666e-01, -1.45705257e+00, -7.83052261e-03])          # JetModel.cpp:13-25
    u2Tnormalization = np.array([108.309, 65.793, 47.333, 31.483])  # JetModel.cpp:26

    # polynomial and partials, standardised units (JetModel.cpp:29-64)
    def compute_f(self, T, Tdot):
        c = self.u2TCoeff
        return c[0] + c[1] * T + c[2] * Tdot + c[3] * T * Tdot + c[4] * T * T + c[5] * Tdot * Tdot

    def compute_g(self, T, Tdot):
        c = self.u2TCoeff
        return c[6] + c[7] * T + c[8] * Tdot + c[9] * T * Tdot + c[10] * T * T + c[11] * Tdot * Tdot

    def compute_df_dT(self, T, Tdot):
        c = self.u2TCoeff
        return c[1] + c[3] * Tdot + 2 * c[4] * T

    def compute_df_dTdot(self, T, Tdot):
        c = self.u2TCoeff
        return c[2] + c[3] * T + 2 * c[5] * Tdot

    def compute_dg_dT(self, T, Tdot):
        c = self.u2TCoeff
        return c[7] + c[9] * Tdot + 2 * c[10] * T

    def compute_dg_dTdot(self, T, Tdot):
        c = self.u2TCoeff
        return c[8] + c[9] * T + 2 * c[11] * Tdot

    def compute_v(self, u):
        return u + self.u2TCoeff[12] * u * u

    # (de)standardisation (JetModel.cpp:66-114)
    def standardizeThrust_u2T(self, thrust):
        return (thrust - self.u2Tnormalization[0]) / self.u2Tnormalization[1]

    def standardizeThrustDot_u2T(self, thrustDot):
        return thrustDot / self.u2Tnormalization[1]

    def standardizeThrottle_u2T(self, throttle):
        return (throttle - self.u2Tnormalization[2]) / self.u2Tnormalization[3]

    def destandardizeThrust_u2T(self, thrustBar):
        return thrustBar * self.u2Tnormalization[1] + self.u2Tnormalization[0]

    def destandardizeThrustDot_u2T(self, thrustDotBar):
        return thrustDotBar * self.u2Tnormalization[1]

    def destandardizeThrottle_u2T(self, v):
        c12 = self.u2TCoeff[12]
        u = (-1.0 + np.sqrt(1.0 + 4.0 * c12 * v)) / (2.0 * c12)
        u = u * self.u2Tnormalization[3] + self.u2Tnormalization[2]
        return np.clip(u, 0.0, 100.0)

    def getThrustStandardDeviation_u2T(self):
        return self.u2Tnormalization[1]

    # convenience used by the workload generator: throttle (percent) that holds thrust T steady
    def steady_state_throttle(self, T):
        Tb = self.standardizeThrust_u2T(T)
        v = -self.compute_f(Tb, 0.0) / self.compute_g(Tb, 0.0)
        return self.destandardizeThrottle_u2T(v)
