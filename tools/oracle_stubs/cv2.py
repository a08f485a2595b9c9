# empty stand-in: the hot path never calls cv2
