#!/usr/bin/env python3
"""numpy model of the device algorithm (layouts, twiddles, polyphase interpolation, SNR identity).

Not product code and not the oracle: a design check that the three-pass structure used by the
HIP kernels (column FFT → fused row FFT·code·IFFT×R → column IFFT + arg-max) equals the
reference's fft/zero-pad/ifft formulation.  Run: python tools/proto_fourstep.py
"""
import numpy as np
rng = np.random.default_rng(0)

def fwd_colpass(x, N1, N2):
    # x[n], n = n1*N2 + n2 ; A[k1][n2] = W_N^{k1 n2} * sum_n1 x[n1][n2] W_N1^{n1 k1}
    X = x.reshape(N1, N2)
    A = np.fft.fft(X, axis=0)
    k1 = np.arange(N1)[:, None]; n2 = np.arange(N2)[None, :]
    return A * np.exp(-2j*np.pi*k1*n2/(N1*N2))

def fwd_rowpass(A):
    # S[k1][k2] = X[k1 + N1*k2]
    return np.fft.fft(A, axis=1)

def inv_rowpass(P, N1, N2):
    # B[k1][q2] = W_N^{-k1 q2} * sum_k2 P[k1][k2] W_N2^{-k2 q2}   (unnormalised)
    B = np.fft.ifft(P, axis=1) * N2
    k1 = np.arange(N1)[:, None]; q2 = np.arange(N2)[None, :]
    return B * np.exp(2j*np.pi*k1*q2/(N1*N2))

def inv_colpass(B, N1):
    # z[q1*N2 + q2] = sum_k1 B[k1][q2] W_N1^{-k1 q1}
    return (np.fft.ifft(B, axis=0) * N1).reshape(-1)

def check(N1, N2, Nint):
    N = N1*N2; R = 2*Nint+1; M = R*N
    code = rng.integers(0, 2, N//2).repeat(2)*2.0-1
    delay = 12345 % N
    y = np.roll(code, delay)*np.exp(1j*0.3) + (rng.standard_normal(N)+1j*rng.standard_normal(N))*2
    fcode = np.conj(np.fft.fft(code))
    # reference formulation
    Y = np.fft.fft(y)
    mul = Y*fcode
    pad = np.zeros(M, complex); pad[:N//2] = mul[:N//2]; pad[-(N//2):] = mul[-(N//2):]
    ref = np.fft.ifft(pad)
    # device formulation
    S = fwd_rowpass(fwd_colpass(y, N1, N2))                   # [k1][k2]
    assert np.allclose(S, Y.reshape(N2, N1).T, atol=1e-6*np.abs(Y).max())
    Cs = fcode.reshape(N2, N1).T                              # same layout
    P = S*Cs
    k2 = np.arange(N2); k2s = np.where(k2 >= N2//2, k2-N2, k2)
    k1 = np.arange(N1)
    z = np.empty(M, complex)
    for rho in range(R):
        ramp = np.exp(2j*np.pi*rho*k1[:, None]/(R*N)) * np.exp(2j*np.pi*rho*k2s[None, :]/(R*N2))
        zr = inv_colpass(inv_rowpass(P*ramp, N1, N2), N1) / M
        z[rho::R] = zr
    err = np.abs(z-ref).max()/np.abs(ref).max()
    ind = int(np.abs(ref).argmax())
    assert int(np.abs(z).argmax()) == ind
    # SNR identity: mean(yincode) = (z[ind-1]+z[ind]+z[ind+1])/M ; mean|yint|^2 = mean|y|^2/R^2
    yint = np.zeros(M, complex); yint[:N//2] = Y[:N//2]; yint[-(N//2):] = Y[-(N//2):]
    yint = np.fft.ifft(yint)
    codetmp = np.repeat(code, R)
    s = ind-1
    yincode = np.concatenate((yint[s:], yint[:s]))*codetmp
    m_direct = yincode.mean()
    m_ident = (ref[(ind-1) % M]+ref[ind]+ref[(ind+1) % M])/M if R == 3 else None
    p_direct = np.mean(np.abs(yincode)**2)
    p_ident = np.mean(np.abs(y)**2)/R**2
    print(f"N1={N1} N2={N2} Nint={Nint}: rel err {err:.2e}  ind {ind} ({ind/R:.2f})",
          f"mean id err {abs(m_direct-m_ident)/abs(m_direct):.2e}" if R == 3 else "",
          f"pow id err {abs(p_direct-p_ident)/p_direct:.2e}")

check(25, 40, 1); check(50, 100, 1); check(100, 200, 0); check(20, 50, 2)
