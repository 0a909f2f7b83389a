// controller_base.hpp -- the serial CPU MPPI controller.
//
// The reference ships `class ControllerBase` (reference include/controller_base.hpp:7-42,
// src/controller_base.cpp:4-98) as an unfinished TensorFlow graph that does not compile; its
// comment block (src/controller_base.cpp:61-80) states the intended pipeline: noise ->
// simulate -> min -> exp -> sum -> div -> weighted mean.  This class keeps the name, the
// constructor signature (k, tau, dt, sDim, aDim) and the method names next / setActions /
// logGraph, takes plain float buffers where the reference took tensorflow::Tensor, and actually
// implements that pipeline -- one thread, one sample after the other, on the per-sample
// PointMassModelGpu / Cost value types (include/point_mass_gpu.hpp, include/cost.hpp).
//
// It is BASELINE config 1 (point_mass1d, K=100, T=50, "plumbing, no GPU") and a CPU controller
// in its own right.  It is NOT a fallback: PointMassModel never routes through it.
// It draws the same Philox noise stream as the GPU engine (same seed -> same E up to the
// last-bit difference between libm and the GPU transcendental units).
#ifndef MPPI_GPU_AMD_CONTROLLER_BASE_HPP_
#define MPPI_GPU_AMD_CONTROLLER_BASE_HPP_

#include <vector>

class ControllerBase {
public:
    ControllerBase(const int k, const int tau, const float dt, const int sDim, const int aDim);
    ~ControllerBase();

    // one MPPI iteration from state x[sDim]: updates the action sequence, writes the action to
    // apply into act[aDim] (may be null) and shifts the sequence.  reference: next(Tensor x)
    void next(const float* x, float* act = nullptr);
    // replace the nominal action sequence, actions[tau*aDim]; false on null.  reference:
    // setActions(vector<Tensor>)
    bool setActions(const float* actions);
    // print the stages of the pipeline (the reference dumped a TF graph); returns 0
    int logGraph() const;

    void setCost(const float* goal, const float* w);                       // each sDim
    void setParams(float lambda, const float* sigma, const float* inv_s);  // null keeps value
    void setSeed(unsigned long long seed);      // restarts the noise stream
    void setNoise(const float* E);              // injected noise [k][tau][aDim]; null = sample
    // worker threads for the sample loops (default 1 = the serial controller).  Samples are
    // independent and every control value is still summed over the samples in order, so the
    // results do not depend on the number of threads.
    void setThreads(int n);

    const std::vector<float>& actions() const { return mU; }
    const std::vector<float>& costs() const { return mCost; }
    const std::vector<float>& weights() const { return mWeights; }
    const std::vector<float>& noise() const { return mE; }
    float beta() const { return mBeta; }
    float nabla() const { return mNabla; }

private:
    int mK, mTau, mSDim, mADim;
    int mThreads;
    float mDt, mLambda;
    unsigned long long mSeed, mSolve;
    bool mInjected;
    std::vector<float> mU, mE, mX, mCost, mWeights, mGoal, mW, mSigma, mInvS;
    float mXGain[4], mUGain[2];
    float mBeta, mNabla;

    void sampleNoise();
};

#endif  // MPPI_GPU_AMD_CONTROLLER_BASE_HPP_
