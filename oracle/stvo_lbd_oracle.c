/*
 * stvo_lbd_oracle.c — TEST INFRASTRUCTURE ONLY: CPU restatement of the LBD line descriptor the reference computes with
 *     Ptr<BinaryDescriptor> lbd = BinaryDescriptor::createBinaryDescriptor();  lbd->compute(img, lines, ldesc);
 * in StereoFrame::detectLineFeatures (/root/reference/src/stereoFrame.cpp:207-243,303) for key-lines that already exist
 * (the LSD / FLD detectors that produce them are NOT restated here — SURVEY.md 8(f) rank 4, second half).
 *
 * Unlike ORB and LSD, LBD's source IS held by the reference: 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp.
 * Every function below cites the lines it follows (paths relative to /root/reference/3rdparty/line_descriptor/src/):
 *   BinaryDescriptor::BinaryDescriptor        binary_descriptor_custom.cpp:217-258   the two Gaussian weight tables (with the
 *                                             source's INTEGER divisions: u = (3 w - 1) / 2 = 10, sigma = (2 w + 1) / 2 = 7,
 *                                             u_g = (9 w - 1) / 2 = 31 for widthOfBand_ w = 7, :109-114)
 *   BinaryDescriptor::computeImpl             :539-687   octave 0 only (the reference detects on one octave,
 *                                             src/stereoFrame.cpp:230), 32 bytes from the 32 band pairs of `combinations` (:74-108)
 *   BinaryDescriptor::computeSobel / computeGaussianPyramid   :350-398   GaussianBlur(5 x 5, sigma 1) then Sobel 3 x 3 to int16
 *   BinaryDescriptor::computeLBD              :1026-1340  the band statistics in FLOAT, in the source's order of operations
 *   BinaryDescriptor::binaryConversion        :401-412
 * PARITY UNPINNED all the same: the file needs OpenCV to compile (cv::Mat, cv::GaussianBlur, cv::Sobel), which this image
 * does not have, and the reference holds no test vector for it.  Third-party arithmetic restated from its published form:
 * cv::GaussianBlur on 8-bit data (OpenCV 3's fixed-point separable filter: kernel x 2^8 rounded, both passes in integers,
 * one rounding shift by 16, BORDER_REFLECT_101 — the same form oracle/stvo_orb_oracle.c uses for ORB's 7 x 7 blur) and
 * cv::Sobel(ksize 3, CV_16S, BORDER_REFLECT_101), which is exact integer arithmetic.
 * Deviations, both defined here and followed bit for bit by the HIP kernels: (1) cos / sin of the FLOAT direction are taken in
 * double precision and rounded to float (the source calls the float overloads of the platform's libm, :1126-1127; the two agree
 * except on rare rounding ties); (2) no fused multiply-adds (the reference's -march=native build may contract a*b+c).
 * The HIP path (stvo-pl_amd/csrc/lbd_kernels.hip) is compared bit for bit with this file (tests/test_gpu_lbd.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LBD_NUM_OF_BANDS 9
#define LBD_WIDTH_OF_BAND 7

static const int kCombinations[32][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6},
                                         {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8},
                                         {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}}; /* :74-108 */

static int reflect101(int p, int n) { /* BORDER_REFLECT_101 */
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

/* cv::GaussianBlur(img, img, Size(5, 5), 1) on 8-bit data (binary_descriptor_custom.cpp:358) */
void orc_gaussian_blur5(const uint8_t* img, int cols, int rows, uint8_t* out) {
    double k[5], sum = 0.0;
    int ki[5];
    for (int i = 0; i < 5; ++i) {
        const double x = i - 2;
        k[i] = exp(-x * x / (2.0 * 1.0 * 1.0));
        sum += k[i];
    }
    for (int i = 0; i < 5; ++i) ki[i] = (int)lrint((float)(k[i] / sum) * 256.0);
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)cols * rows);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int i = 0; i < 5; ++i) s += ki[i] * img[y * cols + reflect101(x + i - 2, cols)];
            tmp[y * cols + x] = s;
        }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int i = 0; i < 5; ++i) s += ki[i] * tmp[reflect101(y + i - 2, rows) * cols + x];
            s = (s + (1 << 15)) >> 16;
            out[y * cols + x] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
        }
    free(tmp);
}

/* cv::Sobel(img, dx, CV_16SC1, 1, 0, 3) and (..., 0, 1, 3)  (binary_descriptor_custom.cpp:395-396): 3 x 3 Sobel, BORDER_REFLECT_101 */
void orc_sobel3(const uint8_t* img, int cols, int rows, int16_t* dx, int16_t* dy) {
    for (int y = 0; y < rows; ++y) {
        const uint8_t* r0 = img + (size_t)reflect101(y - 1, rows) * cols;
        const uint8_t* r1 = img + (size_t)y * cols;
        const uint8_t* r2 = img + (size_t)reflect101(y + 1, rows) * cols;
        for (int x = 0; x < cols; ++x) {
            const int xm = reflect101(x - 1, cols), xp = reflect101(x + 1, cols);
            dx[(size_t)y * cols + x] = (int16_t)((r0[xp] + 2 * r1[xp] + r2[xp]) - (r0[xm] + 2 * r1[xm] + r2[xm]));
            dy[(size_t)y * cols + x] = (int16_t)((r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]));
        }
    }
}

/* the weight tables of the constructor (:225-257): gaussCoefL_[3 w], gaussCoefG_[9 w], kept in double as the source keeps them */
void orc_lbd_tables(double* coefL /* [21] */, double* coefG /* [63] */) {
    const int w = LBD_WIDTH_OF_BAND;
    double u = (w * 3 - 1) / 2;         /* integer division in the source (:228) */
    double sigma = (w * 2 + 1) / 2;     /* integer division in the source (:231) */
    double invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < w * 3; ++i) {
        const double dis = i - u;
        coefL[i] = exp(dis * dis * invsigma2);
    }
    u = (LBD_NUM_OF_BANDS * w - 1) / 2; /* :246 */
    sigma = u;
    invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < LBD_NUM_OF_BANDS * w; ++i) {
        const double dis = i - u;
        coefG[i] = exp(dis * dis * invsigma2);
    }
}

/* computeLBD for one line (:1075-1337).  line = (sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY, direction);
 * desVec receives the 72 floats of the descriptor after both normalisations. */
static void lbd_one_line(const int16_t* pdxImg, const int16_t* pdyImg, int width, int height, const float* line, int numOfPixels,
                         const double* gaussCoefL, const double* gaussCoefG, float* desVec) {
    const int widthOfBand = LBD_WIDTH_OF_BAND;
    const short heightOfLSP = (short)(widthOfBand * LBD_NUM_OF_BANDS);
    const short descriptor_size = LBD_NUM_OF_BANDS * 8;
    float pgdLBandSum[LBD_NUM_OF_BANDS], ngdLBandSum[LBD_NUM_OF_BANDS], pgdL2BandSum[LBD_NUM_OF_BANDS], ngdL2BandSum[LBD_NUM_OF_BANDS];
    float pgdOBandSum[LBD_NUM_OF_BANDS], ngdOBandSum[LBD_NUM_OF_BANDS], pgdO2BandSum[LBD_NUM_OF_BANDS], ngdO2BandSum[LBD_NUM_OF_BANDS];
    memset(pgdLBandSum, 0, sizeof pgdLBandSum); memset(ngdLBandSum, 0, sizeof ngdLBandSum);
    memset(pgdL2BandSum, 0, sizeof pgdL2BandSum); memset(ngdL2BandSum, 0, sizeof ngdL2BandSum);
    memset(pgdOBandSum, 0, sizeof pgdOBandSum); memset(ngdOBandSum, 0, sizeof ngdOBandSum);
    memset(pgdO2BandSum, 0, sizeof pgdO2BandSum); memset(ngdO2BandSum, 0, sizeof ngdO2BandSum);
    const short halfHeight = (short)((heightOfLSP - 1) / 2);
    const short realWidth = (short)width, imageWidth = (short)(realWidth - 1), imageHeight = (short)(height - 1); /* :1101-1103 */
    const short lengthOfLSP = (short)numOfPixels; /* :1117 */
    const short halfWidth = (short)((lengthOfLSP - 1) / 2);
    const float lineMiddlePointX = (float)(0.5 * (line[0] + line[2])); /* :1121-1122 */
    const float lineMiddlePointY = (float)(0.5 * (line[1] + line[3]));
    float dL[2], dO[2];
    dL[0] = (float)cos((double)line[4]); /* :1126-1127 (see the header: double-precision functions rounded to float) */
    dL[1] = (float)sin((double)line[4]);
    dO[0] = -dL[1]; /* :1130-1131 */
    dO[1] = dL[0];
    float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX; /* :1134-1135 */
    float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
    for (short hID = 0; hID < heightOfLSP; hID++) { /* :1138-1226 */
        float sCorX = sCorX0, sCorY = sCorY0;
        float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
        for (short wID = 0; wID < lengthOfLSP; wID++) {
            short tempCor = (short)round(sCorX);
            const short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
            tempCor = (short)round(sCorY);
            const short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
            const short dx = pdxImg[yCor * realWidth + xCor], dy = pdyImg[yCor * realWidth + xCor];
            const float gDL = dx * dL[0] + dy * dL[1];
            const float gDO = dx * dO[0] + dy * dO[1];
            if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
            if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
            sCorX += dL[0];
            sCorY += dL[1];
        }
        sCorX0 -= dL[1];
        sCorY0 += dL[0];
        float coefInGaussion = (float)gaussCoefG[hID];
        pgdLRowSum = coefInGaussion * pgdLRowSum;
        ngdLRowSum = coefInGaussion * ngdLRowSum;
        const float pgdL2RowSum = pgdLRowSum * pgdLRowSum, ngdL2RowSum = ngdLRowSum * ngdLRowSum;
        pgdORowSum = coefInGaussion * pgdORowSum;
        ngdORowSum = coefInGaussion * ngdORowSum;
        const float pgdO2RowSum = pgdORowSum * pgdORowSum, ngdO2RowSum = ngdORowSum * ngdORowSum;
        short bandID = (short)(hID / widthOfBand);
        for (int pass = 0; pass < 3; ++pass) { /* own band, the band above, the band below (:1186-1225) */
            int b, k;
            if (pass == 0) { b = bandID; k = hID % widthOfBand + widthOfBand; }
            else if (pass == 1) { b = bandID - 1; k = hID % widthOfBand + 2 * widthOfBand; if (b < 0) continue; }
            else { b = bandID + 1; k = hID % widthOfBand; if (b >= LBD_NUM_OF_BANDS) continue; }
            coefInGaussion = (float)gaussCoefL[k];
            pgdLBandSum[b] += coefInGaussion * pgdLRowSum;
            ngdLBandSum[b] += coefInGaussion * ngdLRowSum;
            pgdL2BandSum[b] += coefInGaussion * coefInGaussion * pgdL2RowSum;
            ngdL2BandSum[b] += coefInGaussion * coefInGaussion * ngdL2RowSum;
            pgdOBandSum[b] += coefInGaussion * pgdORowSum;
            ngdOBandSum[b] += coefInGaussion * ngdORowSum;
            pgdO2BandSum[b] += coefInGaussion * coefInGaussion * pgdO2RowSum;
            ngdO2BandSum[b] += coefInGaussion * coefInGaussion * ngdO2RowSum;
        }
    }
    /* :1231-1262 */
    const float invN2 = (float)(1.0 / (widthOfBand * 2.0)), invN3 = (float)(1.0 / (widthOfBand * 3.0));
    for (short bandID = 0; bandID < LBD_NUM_OF_BANDS; bandID++) {
        const float invN = (bandID == 0 || bandID == LBD_NUM_OF_BANDS - 1) ? invN2 : invN3;
        const short desID = (short)(bandID * 8);
        float temp = pgdLBandSum[bandID] * invN;
        desVec[desID] = temp;
        desVec[desID + 4] = sqrtf(pgdL2BandSum[bandID] * invN - temp * temp);
        temp = ngdLBandSum[bandID] * invN;
        desVec[desID + 1] = temp;
        desVec[desID + 5] = sqrtf(ngdL2BandSum[bandID] * invN - temp * temp);
        temp = pgdOBandSum[bandID] * invN;
        desVec[desID + 2] = temp;
        desVec[desID + 6] = sqrtf(pgdO2BandSum[bandID] * invN - temp * temp);
        temp = ngdOBandSum[bandID] * invN;
        desVec[desID + 3] = temp;
        desVec[desID + 7] = sqrtf(ngdO2BandSum[bandID] * invN - temp * temp);
    }
    /* normalise means and standard deviations separately (:1265-1298) */
    float tempM = 0, tempS = 0;
    for (int base = 0; base < LBD_NUM_OF_BANDS; ++base) {
        const float* d = desVec + 8 * base;
        tempM += d[0] * d[0]; tempM += d[1] * d[1]; tempM += d[2] * d[2]; tempM += d[3] * d[3];
        tempS += d[4] * d[4]; tempS += d[5] * d[5]; tempS += d[6] * d[6]; tempS += d[7] * d[7];
    }
    tempM = 1 / sqrtf(tempM);
    tempS = 1 / sqrtf(tempS);
    for (int base = 0; base < LBD_NUM_OF_BANDS; ++base) {
        float* d = desVec + 8 * base;
        d[0] = d[0] * tempM; d[1] = d[1] * tempM; d[2] = d[2] * tempM; d[3] = d[3] * tempM;
        d[4] = d[4] * tempS; d[5] = d[5] * tempS; d[6] = d[6] * tempS; d[7] = d[7] * tempS;
    }
    /* clip at 0.4 and re-normalise (:1304-1323) */
    for (short i = 0; i < descriptor_size; i++)
        if (desVec[i] > 0.4) desVec[i] = (float)0.4;
    float temp = 0;
    for (short i = 0; i < descriptor_size; i++) temp += desVec[i] * desVec[i];
    temp = 1 / sqrtf(temp);
    for (short i = 0; i < descriptor_size; i++) desVec[i] = desVec[i] * temp;
}

/* BinaryDescriptor::compute for the octave-0 key-lines of one image.  lines [n][5] = sPointInOctaveX, sPointInOctaveY,
 * ePointInOctaveX, ePointInOctaveY, angle (KeyLine fields as LSDDetectorC / the FLD branch fill them,
 * LSDDetector_custom.cpp:288-298, src/stereoFrame.cpp:272-289); num_pixels [n] = KeyLine::numOfPixels (LineIterator count).
 * desc [n][32]; desc_f (optional) [n][72] the float descriptor. */
void orc_lbd_compute(const uint8_t* img, int cols, int rows, int n, const float* lines, const int32_t* num_pixels, uint8_t* desc,
                     float* desc_f) {
    uint8_t* blur = (uint8_t*)malloc((size_t)cols * rows);
    int16_t* dx = (int16_t*)malloc(sizeof(int16_t) * (size_t)cols * rows);
    int16_t* dy = (int16_t*)malloc(sizeof(int16_t) * (size_t)cols * rows);
    orc_gaussian_blur5(img, cols, rows, blur); /* computeGaussianPyramid :358 */
    orc_sobel3(blur, cols, rows, dx, dy);      /* computeSobel :395-396 */
    double coefL[3 * LBD_WIDTH_OF_BAND], coefG[LBD_NUM_OF_BANDS * LBD_WIDTH_OF_BAND];
    orc_lbd_tables(coefL, coefG);
    for (int l = 0; l < n; ++l) {
        float desVec[LBD_NUM_OF_BANDS * 8];
        lbd_one_line(dx, dy, cols, rows, lines + 5 * l, num_pixels[l], coefL, coefG, desVec);
        if (desc_f) memcpy(desc_f + (size_t)l * LBD_NUM_OF_BANDS * 8, desVec, sizeof desVec);
        for (int comb = 0; comb < 32; ++comb) { /* computeImpl :655-659 + binaryConversion :401-412 */
            const float* f1 = desVec + 8 * kCombinations[comb][0];
            const float* f2 = desVec + 8 * kCombinations[comb][1];
            unsigned result = 0;
            for (int i = 0; i < 8; ++i)
                if (f1[i] > f2[i]) result += 1u << i;
            desc[(size_t)l * 32 + comb] = (uint8_t)result;
        }
    }
    free(dy);
    free(dx);
    free(blur);
}
